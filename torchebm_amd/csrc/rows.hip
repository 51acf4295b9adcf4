// Row-coupled kernels for gfx950: the fused HMC transition kernel (all energies), the
// k-fused Langevin chain for energies whose gradient couples the coordinates of a chain
// (Gaussian, Gaussian mixture), and the stand-alone energy/gradient kernel.
//
// Layout ("lane group per chain"): a chain row x[c, 0:dim] is owned by G consecutive
// lanes of one wavefront (G a power of two, 1..64), each lane holding NV float4 vectors:
// lane lg owns columns (v*G + lg)*4 .. +3 for v < NV.  A wave64 therefore covers 64/G
// whole chains, its global loads/stores of the chain matrix are contiguous 16-byte
// pieces (fully coalesced for dim % 4 == 0), and every per-chain scalar -- potential
// energy, kinetic energy, mixture log-likelihoods, the Metropolis decision -- is a
// cross-lane reduction inside the wavefront (DPP for spans <= 16 lanes, bpermute above),
// never a trip through memory.  The state stays in VGPRs across all MH / Langevin steps;
// LDS holds the shared energy parameters (precision matrix, mixture means) and the
// per-wave exchange buffer the Gaussian mat-vec needs.
#include "ebm_common.h"

namespace ebm {
namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kParamLdsBudget = 56 * 1024;  // bytes of LDS the shared parameters may take

// ---------------------------------------------------------------------------------
// cross-lane helpers
// ---------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

// All-reduce sum over the G lanes of a chain; every lane of the group ends with the
// bit-identical total (each level adds a value to its mirror image).
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if constexpr (G >= 2) v += dpp_f<0xB1>(v);    // quad_perm [1,0,3,2]
  if constexpr (G >= 4) v += dpp_f<0x4E>(v);    // quad_perm [2,3,0,1]
  if constexpr (G >= 8) v += dpp_f<0x141>(v);   // row_half_mirror
  if constexpr (G >= 16) v += dpp_f<0x140>(v);  // row_mirror
  if constexpr (G >= 32) v += __shfl_xor(v, 16);
  if constexpr (G >= 64) v += __shfl_xor(v, 32);
  return v;
}

template <int G>
__device__ __forceinline__ bool group_any(bool flag) {
  if constexpr (G == 1) return flag;
  const unsigned long long b = __ballot(flag);
  if constexpr (G == 64) return b != 0ull;
  const int lane = threadIdx.x & 63;
  const unsigned long long m = ((1ull << G) - 1ull) << (lane & ~(G - 1));
  return (b & m) != 0ull;
}

// ---------------------------------------------------------------------------------
// geometry of one lane
// ---------------------------------------------------------------------------------
template <int G, int NV>
struct Lane {
  int64_t chain;      // chain row owned by this lane's group
  int lg;             // lane index inside the group
  int chain_in_wave;  // 0 .. 64/G-1
  int wave;           // wave index inside the block
  bool active;        // chain < n_chains
  bool vec_ok;        // dim % 4 == 0: float4 global accesses are aligned
  int dim;
  int col[NV];        // first column of vector v
  unsigned valid;     // bit (v*4+i): column col[v]+i < dim and the chain is active

  __device__ __forceinline__ void init(int64_t n_chains, int dim_) {
    const int64_t tid = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    chain = tid / G;
    lg = (int)(tid % G);
    chain_in_wave = (threadIdx.x & 63) / G;
    wave = threadIdx.x >> 6;
    active = chain < n_chains;
    dim = dim_;
    vec_ok = (dim_ & 3) == 0;
    valid = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      col[v] = (v * G + lg) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (active && col[v] + i < dim_) valid |= 1u << (v * 4 + i);
    }
  }
  __device__ __forceinline__ bool ok(int v, int i) const { return (valid >> (v * 4 + i)) & 1u; }
  __device__ __forceinline__ bool full(int v) const { return ((valid >> (v * 4)) & 0xFu) == 0xFu; }
};

template <int NV>
struct Slice {
  float a[NV][4];
};

template <int G, int NV>
__device__ __forceinline__ void load_slice(const Lane<G, NV>& L, const float* __restrict__ base,
                                           int64_t row_off, Slice<NV>& s) {
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    if (L.vec_ok && L.full(v)) {
      const float4 t = *reinterpret_cast<const float4*>(base + row_off + L.col[v]);
      s.a[v][0] = t.x; s.a[v][1] = t.y; s.a[v][2] = t.z; s.a[v][3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) s.a[v][i] = L.ok(v, i) ? base[row_off + L.col[v] + i] : 0.0f;
    }
  }
}

template <int G, int NV>
__device__ __forceinline__ void store_slice(const Lane<G, NV>& L, float* __restrict__ base,
                                            int64_t row_off, const Slice<NV>& s) {
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    if (L.vec_ok && L.full(v)) {
      *reinterpret_cast<float4*>(base + row_off + L.col[v]) =
          make_float4(s.a[v][0], s.a[v][1], s.a[v][2], s.a[v][3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (L.ok(v, i)) base[row_off + L.col[v] + i] = s.a[v][i];
    }
  }
}

// Load a [dim] parameter vector slice (mean, diagonal mass); `fill` in invalid slots.
template <int G, int NV>
__device__ __forceinline__ void load_param_slice(const Lane<G, NV>& L, const float* __restrict__ p,
                                                 float fill, Slice<NV>& s) {
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int i = 0; i < 4; ++i) s.a[v][i] = (L.col[v] + i < L.dim) ? p[L.col[v] + i] : fill;
}

// Native-RNG normals for this lane's slice at `step` (flat element e = chain*dim + col).
template <int G, int NV>
__device__ __forceinline__ void normal_slice(const Lane<G, NV>& L, RngKey key, uint64_t step,
                                             Slice<NV>& s) {
  const uint64_t row0 = (uint64_t)L.chain * (uint64_t)L.dim;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    if (L.vec_ok) {  // e % 4 == 0: the slice vector is exactly one Philox counter
      const F4 n = normal4_at(key, (row0 + (uint64_t)L.col[v]) >> 2, step);
#pragma unroll
      for (int i = 0; i < 4; ++i) s.a[v][i] = n.v[i];
    } else {
      uint64_t have = ~0ull;
      F4 n;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint64_t e = row0 + (uint64_t)(L.col[v] + i);
        if ((e >> 2) != have) {
          have = e >> 2;
          n = normal4_at(key, have, step);
        }
        const int r = (int)(e & 3);
        s.a[v][i] = r == 0 ? n.v[0] : (r == 1 ? n.v[1] : (r == 2 ? n.v[2] : n.v[3]));
      }
    }
  }
}

// ---------------------------------------------------------------------------------
// energies: E(x) (group-reduced, identical on every lane of the group) and dE/dx for
// this lane's slice.  Invalid slots hold x = 0 and must yield g = 0 and no energy.
// ---------------------------------------------------------------------------------
struct EnergyParams {
  int kind;
  int n_comp;
  float s0, s1;
  const float* dev0;  // global
  const float* dev1;
  int param_in_lds;   // shared parameters were staged into LDS
  int dim_pad;        // row stride of the staged parameters (dim rounded up to 4)
};

// LDS carve-up (dynamic shared memory, 16-byte aligned):
//   [0, param_floats)                      shared parameters (P rows / mixture means + log-weights)
//   [param_floats, + waves * xchg_floats)  per-wave exchange rows (Gaussian only)
struct Smem {
  float* param;
  float* xchg;  // this wave's exchange buffer
};

template <int KIND, int G, int NV>
struct Energy;

template <int G, int NV>
struct Energy<EBM_ENERGY_DOUBLE_WELL, G, NV> {
  float h, b2;
  __device__ __forceinline__ void init(const EnergyParams& P, const Lane<G, NV>&, const Smem&) {
    h = P.s0; b2 = P.s1;
  }
  template <bool WANT_E>
  __device__ __forceinline__ float eval(const Lane<G, NV>& L, const Slice<NV>& x, Slice<NV>& g) const {
    float acc = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xv = x.a[v][i];
        const float u = xv * xv - b2;
        const bool ok = L.ok(v, i);
        g.a[v][i] = ok ? (h * (2.0f * u)) * (2.0f * xv) : 0.0f;
        if (WANT_E) acc += ok ? u * u : 0.0f;
      }
    if (!WANT_E) return 0.0f;
    return h * group_sum<G>(acc);
  }
};

template <int G, int NV>
struct Energy<EBM_ENERGY_HARMONIC, G, NV> {
  float hk;
  __device__ __forceinline__ void init(const EnergyParams& P, const Lane<G, NV>&, const Smem&) { hk = P.s0; }
  template <bool WANT_E>
  __device__ __forceinline__ float eval(const Lane<G, NV>&, const Slice<NV>& x, Slice<NV>& g) const {
    float acc = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xv = x.a[v][i];  // invalid slots are 0 and contribute 0
        g.a[v][i] = hk * (2.0f * xv);
        if (WANT_E) acc += xv * xv;
      }
    if (!WANT_E) return 0.0f;
    return hk * group_sum<G>(acc);
  }
};

// Gaussian: g = Ps d with Ps = (P + P^T)/2 (what autograd returns for 0.5 d^T P d), staged
// row-major in LDS; d is exchanged through the wave's LDS row so that every lane can walk
// all dim coordinates of its chain.  Ps symmetric => column slice of row j == needed block.
template <int G, int NV>
struct Energy<EBM_ENERGY_GAUSSIAN, G, NV> {
  Slice<NV> mu;
  const float* P_lds;
  const float* P_glb;
  float* xrow;  // this chain's exchange row in LDS
  int dim_pad;
  __device__ __forceinline__ void init(const EnergyParams& P, const Lane<G, NV>& L, const Smem& S) {
    load_param_slice(L, P.dev0, 0.0f, mu);
    P_lds = P.param_in_lds ? S.param : nullptr;
    P_glb = P.dev1;
    dim_pad = P.dim_pad;
    xrow = S.xchg + L.chain_in_wave * (G * NV * 4);
  }
  template <bool WANT_E>
  __device__ __forceinline__ float eval(const Lane<G, NV>& L, const Slice<NV>& x, Slice<NV>& g) const {
    Slice<NV> d;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        d.a[v][i] = L.ok(v, i) ? x.a[v][i] - mu.a[v][i] : 0.0f;
        g.a[v][i] = 0.0f;
      }
      *reinterpret_cast<float4*>(xrow + L.col[v]) = make_float4(d.a[v][0], d.a[v][1], d.a[v][2], d.a[v][3]);
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (P_lds) {
      for (int j = 0; j < L.dim; ++j) {
        const float dj = xrow[j];
        const float* row = P_lds + j * dim_pad;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float4 pr = *reinterpret_cast<const float4*>(row + L.col[v]);
          g.a[v][0] = __builtin_fmaf(pr.x, dj, g.a[v][0]);
          g.a[v][1] = __builtin_fmaf(pr.y, dj, g.a[v][1]);
          g.a[v][2] = __builtin_fmaf(pr.z, dj, g.a[v][2]);
          g.a[v][3] = __builtin_fmaf(pr.w, dj, g.a[v][3]);
        }
      }
    } else {  // precision matrix too large for LDS: stream rows from L2
      for (int j = 0; j < L.dim; ++j) {
        const float dj = xrow[j];
        const float* row = P_glb + (int64_t)j * L.dim;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (L.col[v] + i < L.dim) g.a[v][i] = __builtin_fmaf(row[L.col[v] + i], dj, g.a[v][i]);
      }
    }
    __builtin_amdgcn_wave_barrier();
    float acc = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!L.ok(v, i)) g.a[v][i] = 0.0f;
        if (WANT_E) acc = __builtin_fmaf(d.a[v][i], g.a[v][i], acc);
      }
    if (!WANT_E) return 0.0f;
    return 0.5f * group_sum<G>(acc);
  }
};

// Gaussian mixture (isotropic, shared sigma): one pass over the K components with a
// running max (online softmax).  g = s1 * (x - sum_k r_k mu_k), E = -(m + log sum_k e^{l_k-m}).
template <int G, int NV>
struct Energy<EBM_ENERGY_GMM, G, NV> {
  const float* mu_lds;
  const float* mu_glb;
  const float* logw;
  int K, dim_pad;
  float inv2s2, invs2;
  __device__ __forceinline__ void init(const EnergyParams& P, const Lane<G, NV>&, const Smem& S) {
    mu_lds = P.param_in_lds ? S.param : nullptr;
    mu_glb = P.dev0;
    logw = P.param_in_lds ? S.param + P.n_comp * P.dim_pad : P.dev1;
    K = P.n_comp;
    dim_pad = P.dim_pad;
    inv2s2 = P.s0;
    invs2 = P.s1;
  }
  __device__ __forceinline__ void load_mu(const Lane<G, NV>& L, int k, int v, float (&m)[4]) const {
    if (mu_lds) {
      const float4 t = *reinterpret_cast<const float4*>(mu_lds + k * dim_pad + L.col[v]);
      m[0] = t.x; m[1] = t.y; m[2] = t.z; m[3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        m[i] = (L.col[v] + i < L.dim) ? mu_glb[(int64_t)k * L.dim + L.col[v] + i] : 0.0f;
    }
  }
  template <bool WANT_E>
  __device__ __forceinline__ float eval(const Lane<G, NV>& L, const Slice<NV>& x, Slice<NV>& g) const {
    float run_max = -__builtin_inff();
    float run_sum = 0.0f;
    Slice<NV> acc;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc.a[v][i] = 0.0f;
    for (int k = 0; k < K; ++k) {
      float mk[NV][4];
      float dist = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        load_mu(L, k, v, mk[v]);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float df = L.ok(v, i) ? x.a[v][i] - mk[v][i] : 0.0f;
          dist = __builtin_fmaf(df, df, dist);
        }
      }
      dist = group_sum<G>(dist);
      const float logit = __builtin_fmaf(-dist, inv2s2, logw[k]);
      const float new_max = logit > run_max ? logit : run_max;
      const float scale = __expf(run_max - new_max);  // 0 on the first component
      const float w = __expf(logit - new_max);
      run_sum = __builtin_fmaf(run_sum, scale, w);
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc.a[v][i] = __builtin_fmaf(w, mk[v][i], acc.a[v][i] * scale);
      run_max = new_max;
    }
    const float inv = 1.0f / run_sum;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        g.a[v][i] = L.ok(v, i) ? invs2 * (x.a[v][i] - acc.a[v][i] * inv) : 0.0f;
    if (!WANT_E) return 0.0f;
    return -(run_max + logf(run_sum));
  }
};

// Stage the shared parameters into LDS (all threads of the block), zero-padded rows.
__device__ __forceinline__ void stage_params(const EnergyParams& P, int dim, float* dst) {
  if (!P.param_in_lds) return;
  if (P.kind == EBM_ENERGY_GAUSSIAN) {
    const int n = dim * P.dim_pad;
    for (int i = threadIdx.x; i < n; i += kBlock) {
      const int r = i / P.dim_pad, c = i - r * P.dim_pad;
      dst[i] = (c < dim) ? P.dev1[(int64_t)r * dim + c] : 0.0f;
    }
  } else if (P.kind == EBM_ENERGY_GMM) {
    const int n = P.n_comp * P.dim_pad;
    for (int i = threadIdx.x; i < n; i += kBlock) {
      const int r = i / P.dim_pad, c = i - r * P.dim_pad;
      dst[i] = (c < dim) ? P.dev0[(int64_t)r * dim + c] : 0.0f;
    }
    for (int i = threadIdx.x; i < P.n_comp; i += kBlock) dst[n + i] = P.dev1[i];
  }
  __syncthreads();
}

extern __shared__ __attribute__((aligned(16))) float ebm_smem[];

template <int G, int NV>
__device__ __forceinline__ Smem carve_smem(int param_floats) {
  Smem S;
  S.param = ebm_smem;
  S.xchg = ebm_smem + param_floats + (threadIdx.x >> 6) * (64 * NV * 4);
  return S;
}

// ---------------------------------------------------------------------------------
// HMC transition kernel
// ---------------------------------------------------------------------------------
struct HmcArgs {
  float* x;
  int64_t n_chains;
  int32_t dim;
  int32_t n_mh;
  int32_t n_leapfrog;
  float eps;
  const float* eps_table;
  int32_t mass_kind;
  float mass_raw, mass_sqrt, mass_safe;  // scalar mass forms
  const float* mass_diag;
  int32_t thin;
  int32_t n_kept;
  float* traj;
  uint8_t* accept_mask;
  uint32_t* accept_count;
  const float* p_noise;
  const float* u;
  RngKey key;
  uint64_t step0;
  EnergyParams energy;
  int param_floats;
};

template <int KIND, int G, int NV>
__global__ __launch_bounds__(kBlock) void hmc_chain_kernel(HmcArgs a) {
  Lane<G, NV> L;
  L.init(a.n_chains, a.dim);
  const Smem S = carve_smem<G, NV>(a.param_floats);
  stage_params(a.energy, a.dim, S.param);
  Energy<KIND, G, NV> en;
  en.init(a.energy, L, S);

  const int64_t row = L.active ? L.chain * (int64_t)a.dim : 0;
  Slice<NV> xc;  // current (accepted) state
  load_slice(L, a.x, row, xc);

  // diagonal mass: raw (kinetic), sqrt (momentum draw), clamped (drift)
  Slice<NV> m_raw, m_sqrt, m_safe;
  if (a.mass_kind == EBM_MASS_DIAG) {
    load_param_slice(L, a.mass_diag, 1.0f, m_raw);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        m_sqrt.a[v][i] = sqrtf(m_raw.a[v][i]);
        m_safe.a[v][i] = m_raw.a[v][i] < 1e-10f ? 1e-10f : m_raw.a[v][i];
      }
  }

  const int64_t traj_row = L.active ? L.chain * (int64_t)a.n_kept * a.dim : 0;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  float eps = a.eps;

  for (int t = 0; t < a.n_mh; ++t) {
    if (a.eps_table) eps = a.eps_table[t];
    const float half_eps = 0.5f * eps;

    // ---- momentum draw: p ~ N(0, M)  (samplers/hmc.py:92-134)
    Slice<NV> p;
    if (a.p_noise) load_slice(L, a.p_noise, ((int64_t)t * a.n_chains) * a.dim + row, p);
    else normal_slice(L, a.key, a.step0 + 2ull * (uint64_t)t, p);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (!L.ok(v, i)) p.a[v][i] = 0.0f;
        else if (a.mass_kind == EBM_MASS_SCALAR) p.a[v][i] = p.a[v][i] * a.mass_sqrt;
        else if (a.mass_kind == EBM_MASS_DIAG) p.a[v][i] = p.a[v][i] * m_sqrt.a[v][i];
      }

    auto kinetic = [&](const Slice<NV>& q) -> float {
      float acc = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float sq = q.a[v][i] * q.a[v][i];
          if (a.mass_kind == EBM_MASS_DIAG) sq = sq / m_raw.a[v][i];
          acc += L.ok(v, i) ? sq : 0.0f;
        }
      float k = 0.5f * group_sum<G>(acc);
      if (a.mass_kind == EBM_MASS_SCALAR) k = k / a.mass_raw;
      return clamp_nanprop(k, 0.0f, 1e10f);
    };

    // ---- H0 and the first force
    Slice<NV> g;
    float e0 = en.template eval<true>(L, xc, g);
    const float h0 = clamp_nanprop(e0, -1e10f, 1e10f) + kinetic(p);

    // ---- L leapfrog steps, safe mode (integrators/leapfrog.py:156-185); the force at
    //      the end of a step is bit-identical to the reference's recomputed force at the
    //      start of the next one, so it is reused unless the NaN scrub changed x.
    Slice<NV> x = xc;
    float e1 = e0;
    for (int l = 0; l < a.n_leapfrog; ++l) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);
          const float ph = p.a[v][i] + half_eps * f;
          float step = eps * ph;
          if (a.mass_kind == EBM_MASS_SCALAR) step = step / a.mass_safe;
          else if (a.mass_kind == EBM_MASS_DIAG) step = step / m_safe.a[v][i];
          p.a[v][i] = ph;
          x.a[v][i] = L.ok(v, i) ? x.a[v][i] + step : 0.0f;
        }
      e1 = en.template eval<true>(L, x, g);
      bool changed = false;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float f = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);
          const float pn = nan_to_num0(p.a[v][i] + half_eps * f);
          const float xs = nan_to_num0(x.a[v][i]);
          changed |= !(xs == x.a[v][i]);
          p.a[v][i] = L.ok(v, i) ? pn : 0.0f;
          x.a[v][i] = xs;
        }
      if (group_any<G>(changed)) e1 = en.template eval<true>(L, x, g);
    }
    const float h1 = clamp_nanprop(e1, -1e10f, 1e10f) + kinetic(p);

    // ---- Metropolis accept (samplers/hmc.py:277-292)
    const float dlt = clamp_nanprop(h0 - h1, -50.0f, 50.0f);
    float acc_p = expf(dlt);
    acc_p = (acc_p > 1.0f) ? 1.0f : acc_p;
    float uu;
    if (a.u) uu = L.active ? a.u[(int64_t)t * a.n_chains + L.chain] : 2.0f;
    else uu = u01_half_open(pick(philox_at(a.key, (uint64_t)L.chain >> 2, a.step0 + 2ull * (uint64_t)t + 1ull),
                                 (int)(L.chain & 3)));
    const bool accept = L.active && (uu < acc_p);
    if (accept) xc = x;

    const bool leader = L.active && L.lg == 0;
    if (a.accept_mask && leader) a.accept_mask[(int64_t)t * a.n_chains + L.chain] = accept ? 1 : 0;
    if (a.accept_count) {  // wavefront-level count, one atomic per wave
      const unsigned long long b = __ballot(accept && leader);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(a.accept_count + t, (uint32_t)__popcll(b));
    }

    if (a.traj && --until_keep == 0) {
      until_keep = a.thin;
      store_slice(L, a.traj, traj_row + keep_off, xc);
      keep_off += a.dim;
    }
  }
  store_slice(L, a.x, row, xc);
}

// ---------------------------------------------------------------------------------
// Langevin chain for row-coupled energies
// ---------------------------------------------------------------------------------
struct RowChainArgs {
  float* x;
  int64_t n_chains;
  int32_t dim;
  int32_t k_steps;
  float eta, sqrt_eta, noise_coef;
  const float4* table;
  int clamp_on;
  float cmin, cmax;
  int32_t thin, n_kept;
  float* traj;
  const float* noise;
  RngKey key;
  uint64_t step0;
  EnergyParams energy;
  int param_floats;
};

template <int KIND, int G, int NV>
__global__ __launch_bounds__(kBlock) void langevin_chain_rows_kernel(RowChainArgs a) {
  Lane<G, NV> L;
  L.init(a.n_chains, a.dim);
  const Smem S = carve_smem<G, NV>(a.param_floats);
  stage_params(a.energy, a.dim, S.param);
  Energy<KIND, G, NV> en;
  en.init(a.energy, L, S);

  const int64_t row = L.active ? L.chain * (int64_t)a.dim : 0;
  Slice<NV> x;
  load_slice(L, a.x, row, x);
  const int64_t traj_row = L.active ? L.chain * (int64_t)a.n_kept * a.dim : 0;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;

  for (int s = 0; s < a.k_steps; ++s) {
    if (a.table) {
      const float4 t = a.table[s];
      eta = t.x; sqrt_eta = t.y; noise_coef = t.z;
    }
    Slice<NV> g, eps;
    en.template eval<false>(L, x, g);
    if (a.noise) load_slice(L, a.noise, ((int64_t)s * a.n_chains) * a.dim + row, eps);
    else normal_slice(L, a.key, a.step0 + (uint64_t)s, eps);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x1 = x.a[v][i] - eta * g.a[v][i];
        const float dw = eps.a[v][i] * sqrt_eta;
        float nv = x1 + noise_coef * dw;
        if (a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
        x.a[v][i] = L.ok(v, i) ? nv : 0.0f;
      }
    if (a.traj && --until_keep == 0) {
      until_keep = a.thin;
      store_slice(L, a.traj, traj_row + keep_off, x);
      keep_off += a.dim;
    }
  }
  store_slice(L, a.x, row, x);
}

// ---------------------------------------------------------------------------------
// energy + gradient
// ---------------------------------------------------------------------------------
struct EgArgs {
  const float* x;
  int64_t n_chains;
  int32_t dim;
  float* e_out;
  float* g_out;
  EnergyParams energy;
  int param_floats;
};

template <int KIND, int G, int NV>
__global__ __launch_bounds__(kBlock) void energy_grad_kernel(EgArgs a) {
  Lane<G, NV> L;
  L.init(a.n_chains, a.dim);
  const Smem S = carve_smem<G, NV>(a.param_floats);
  stage_params(a.energy, a.dim, S.param);
  Energy<KIND, G, NV> en;
  en.init(a.energy, L, S);
  const int64_t row = L.active ? L.chain * (int64_t)a.dim : 0;
  Slice<NV> x, g;
  load_slice(L, a.x, row, x);
  const float e = en.template eval<true>(L, x, g);
  if (a.e_out && L.active && L.lg == 0) a.e_out[L.chain] = e;
  if (a.g_out) store_slice(L, a.g_out, row, g);
}

// ---------------------------------------------------------------------------------
// host side: geometry selection and dispatch
// ---------------------------------------------------------------------------------
struct Geometry {
  int G, NV;
};

bool pick_geometry(int dim, Geometry& geo) {
  const int nvec = (dim + 3) / 4;
  if (nvec <= 64) {
    int g = 1;
    while (g < nvec) g <<= 1;
    geo = Geometry{g, 1};
    return true;
  }
  if (nvec <= 128) { geo = Geometry{64, 2}; return true; }
  if (nvec <= 256) { geo = Geometry{64, 4}; return true; }
  return false;
}

// Decide where the shared parameters live and how much dynamic LDS the launch needs.
void plan_params(const ebm_energy_t& e, int dim, const Geometry& geo, EnergyParams& P,
                 int& param_floats, size_t& smem_bytes) {
  P.kind = e.kind; P.n_comp = e.n_comp; P.s0 = e.s[0]; P.s1 = e.s[1];
  P.dev0 = e.dev0; P.dev1 = e.dev1;
  P.dim_pad = (dim + 3) & ~3;
  P.param_in_lds = 0;
  param_floats = 0;
  size_t xchg = 0;
  if (e.kind == EBM_ENERGY_GAUSSIAN) {
    const size_t need = (size_t)dim * P.dim_pad;
    if (need * 4 <= (size_t)kParamLdsBudget) { P.param_in_lds = 1; param_floats = (int)need; }
    xchg = (size_t)kWavesPerBlock * 64 * geo.NV * 4;
  } else if (e.kind == EBM_ENERGY_GMM) {
    const size_t need = (size_t)e.n_comp * P.dim_pad + (size_t)((e.n_comp + 3) & ~3);
    if (need * 4 <= (size_t)kParamLdsBudget) { P.param_in_lds = 1; param_floats = (int)need; }
  }
  smem_bytes = ((size_t)param_floats + xchg) * sizeof(float);
}

#define EBM_GEO_SWITCH(KERNEL, KIND, geo, ...)                                              \
  do {                                                                                      \
    if (geo.NV == 1) {                                                                      \
      switch (geo.G) {                                                                      \
        case 1:  hipLaunchKernelGGL((KERNEL<KIND, 1, 1>), __VA_ARGS__); break;              \
        case 2:  hipLaunchKernelGGL((KERNEL<KIND, 2, 1>), __VA_ARGS__); break;              \
        case 4:  hipLaunchKernelGGL((KERNEL<KIND, 4, 1>), __VA_ARGS__); break;              \
        case 8:  hipLaunchKernelGGL((KERNEL<KIND, 8, 1>), __VA_ARGS__); break;              \
        case 16: hipLaunchKernelGGL((KERNEL<KIND, 16, 1>), __VA_ARGS__); break;             \
        case 32: hipLaunchKernelGGL((KERNEL<KIND, 32, 1>), __VA_ARGS__); break;             \
        default: hipLaunchKernelGGL((KERNEL<KIND, 64, 1>), __VA_ARGS__); break;             \
      }                                                                                     \
    } else if (geo.NV == 2) {                                                               \
      hipLaunchKernelGGL((KERNEL<KIND, 64, 2>), __VA_ARGS__);                               \
    } else {                                                                                \
      hipLaunchKernelGGL((KERNEL<KIND, 64, 4>), __VA_ARGS__);                               \
    }                                                                                       \
  } while (0)

#define EBM_KIND_SWITCH(KERNEL, kind, geo, ...)                                             \
  do {                                                                                      \
    switch (kind) {                                                                         \
      case EBM_ENERGY_DOUBLE_WELL: EBM_GEO_SWITCH(KERNEL, EBM_ENERGY_DOUBLE_WELL, geo, __VA_ARGS__); break; \
      case EBM_ENERGY_HARMONIC:    EBM_GEO_SWITCH(KERNEL, EBM_ENERGY_HARMONIC, geo, __VA_ARGS__); break;    \
      case EBM_ENERGY_GAUSSIAN:    EBM_GEO_SWITCH(KERNEL, EBM_ENERGY_GAUSSIAN, geo, __VA_ARGS__); break;    \
      default:                     EBM_GEO_SWITCH(KERNEL, EBM_ENERGY_GMM, geo, __VA_ARGS__); break;         \
    }                                                                                       \
  } while (0)

int64_t blocks_for(int64_t n_chains, const Geometry& geo) {
  const int chains_per_block = kBlock / geo.G;
  return ceil_div64(n_chains, chains_per_block);
}

}  // namespace

int launch_hmc_chain(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh,
                     int32_t n_leapfrog, float eps, const float* eps_table, int32_t mass_kind,
                     double mass_scalar, const float* mass_diag, int32_t thin, float* traj,
                     uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise,
                     const float* u, uint64_t seed, uint64_t offset, hipStream_t st) {
  Geometry geo;
  if (!pick_geometry(dim, geo)) return fail(EBM_EDIM, "ebm_hmc_chain_f32: dim %d > 1024 is not supported by the fused kernel", dim);
  HmcArgs a;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.n_mh = n_mh; a.n_leapfrog = n_leapfrog;
  a.eps = eps; a.eps_table = eps_table; a.mass_kind = mass_kind;
  a.mass_raw = (float)mass_scalar;
  a.mass_sqrt = (float)sqrt(mass_scalar);
  a.mass_safe = (float)(mass_scalar < 1e-10 ? 1e-10 : mass_scalar);
  a.mass_diag = mass_diag; a.thin = thin; a.n_kept = n_mh / thin; a.traj = traj;
  a.accept_mask = accept_mask; a.accept_count = accept_count; a.p_noise = p_noise; a.u = u;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  size_t smem = 0;
  plan_params(e, dim, geo, a.energy, a.param_floats, smem);
  const int64_t blocks = blocks_for(n_chains, geo);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_hmc_chain_f32: too many chains for one launch");
  EBM_KIND_SWITCH(hmc_chain_kernel, e.kind, geo, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_hmc_chain_f32");
}

int launch_langevin_chain_rows(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim,
                               int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                               const float* coef_table, int clamp_on, float cmin, float cmax,
                               int32_t thin, float* traj, const float* noise, uint64_t seed,
                               uint64_t offset, hipStream_t st) {
  Geometry geo;
  if (!pick_geometry(dim, geo)) return fail(EBM_EDIM, "ebm_langevin_chain_f32: dim %d > 1024 is not supported for this energy", dim);
  RowChainArgs a;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  size_t smem = 0;
  plan_params(e, dim, geo, a.energy, a.param_floats, smem);
  const int64_t blocks = blocks_for(n_chains, geo);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  if (e.kind == EBM_ENERGY_GAUSSIAN)
    EBM_GEO_SWITCH(langevin_chain_rows_kernel, EBM_ENERGY_GAUSSIAN, geo, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else
    EBM_GEO_SWITCH(langevin_chain_rows_kernel, EBM_ENERGY_GMM, geo, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_langevin_chain_f32");
}

int launch_energy_grad(const ebm_energy_t& e, const float* x, int64_t n_chains, int32_t dim,
                       float* e_out, float* g_out, hipStream_t st) {
  Geometry geo;
  if (!pick_geometry(dim, geo)) return fail(EBM_EDIM, "ebm_energy_grad_f32: dim %d > 1024 is not supported", dim);
  EgArgs a;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.e_out = e_out; a.g_out = g_out;
  size_t smem = 0;
  plan_params(e, dim, geo, a.energy, a.param_floats, smem);
  const int64_t blocks = blocks_for(n_chains, geo);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_energy_grad_f32: too many chains for one launch");
  EBM_KIND_SWITCH(energy_grad_kernel, e.kind, geo, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_energy_grad_f32");
}

}  // namespace ebm
