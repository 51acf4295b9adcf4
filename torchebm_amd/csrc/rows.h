// Shared device templates of the row-coupled kernels (hmc.hip, rows_langevin.hip).
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
//
// FULL = (dim == 4*G*NV): every slot of every lane is a real column, so the per-slot
// validity selects disappear from the inner loops; lanes of chains past n_chains then
// compute on zeros and are only prevented from loading/storing.
#pragma once
#include "diag.h"
#include "ebm_common.h"

namespace ebm {
namespace rows {

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
  const unsigned long long m = ((1ull << (G & 63)) - 1ull) << (lane & ~(G - 1));
  return (b & m) != 0ull;
}

// ---------------------------------------------------------------------------------
// geometry of one lane
// ---------------------------------------------------------------------------------
template <int G_, int NV_, bool FULL_>
struct Lane {
  static constexpr int G = G_;
  static constexpr int NV = NV_;
  static constexpr bool FULL = FULL_;

  int64_t chain;      // chain row owned by this lane's group
  int lg;             // lane index inside the group
  int chain_in_wave;  // 0 .. 64/G-1
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
    active = chain < n_chains;
    dim = dim_;
    vec_ok = FULL || (dim_ & 3) == 0;
    valid = 0;
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      col[v] = (v * G + lg) * 4;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (active && col[v] + i < dim_) valid |= 1u << (v * 4 + i);
    }
  }
  // "this slot is a real element that takes part in the arithmetic"
  __device__ __forceinline__ bool ok(int v, int i) const {
    if constexpr (FULL) return true;
    return (valid >> (v * 4 + i)) & 1u;
  }
  // "this slot may be loaded from / stored to memory"
  __device__ __forceinline__ bool mem_ok(int v, int i) const { return (valid >> (v * 4 + i)) & 1u; }
  __device__ __forceinline__ bool mem_full(int v) const { return ((valid >> (v * 4)) & 0xFu) == 0xFu; }
  __device__ __forceinline__ bool col_ok(int v, int i) const {
    if constexpr (FULL) return true;
    return col[v] + i < dim;
  }
};

template <int NV>
struct Slice {
  float a[NV][4];
};

template <class LaneT>
__device__ __forceinline__ void load_slice(const LaneT& L, const float* __restrict__ base,
                                           int64_t row_off, Slice<LaneT::NV>& s) {
#pragma unroll
  for (int v = 0; v < LaneT::NV; ++v) {
    if (L.vec_ok && L.mem_full(v)) {
      const float4 t = *reinterpret_cast<const float4*>(base + row_off + L.col[v]);
      s.a[v][0] = t.x; s.a[v][1] = t.y; s.a[v][2] = t.z; s.a[v][3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) s.a[v][i] = L.mem_ok(v, i) ? base[row_off + L.col[v] + i] : 0.0f;
    }
  }
}

template <class LaneT>
__device__ __forceinline__ void store_slice(const LaneT& L, float* __restrict__ base,
                                            int64_t row_off, const Slice<LaneT::NV>& s) {
#pragma unroll
  for (int v = 0; v < LaneT::NV; ++v) {
    if (L.vec_ok && L.mem_full(v)) {
      *reinterpret_cast<float4*>(base + row_off + L.col[v]) =
          make_float4(s.a[v][0], s.a[v][1], s.a[v][2], s.a[v][3]);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        if (L.mem_ok(v, i)) base[row_off + L.col[v] + i] = s.a[v][i];
    }
  }
}

// Diagnostics tile (diag.h): the block's chains in flat order, chain_in_block * dim + column.
template <class LaneT>
__device__ __forceinline__ void tile_store(const LaneT& L, float* __restrict__ tile, const Slice<LaneT::NV>& s) {
  float* row = tile + (int)(threadIdx.x / LaneT::G) * L.dim;
#pragma unroll
  for (int v = 0; v < LaneT::NV; ++v)
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (L.mem_ok(v, i)) row[L.col[v] + i] = s.a[v][i];
}

// valid flat elements of this workgroup's tile: its active chains x dim
template <int G>
__device__ __forceinline__ int tile_valid(int64_t n_chains, int dim) {
  const int64_t first = (int64_t)blockIdx.x * (kBlock / G);
  const int64_t left = n_chains - first;
  const int c = left >= kBlock / G ? kBlock / G : (left > 0 ? (int)left : 0);
  return c * dim;
}

// Load a [dim] parameter vector slice (mean, diagonal mass); `fill` in invalid slots.
template <class LaneT>
__device__ __forceinline__ void load_param_slice(const LaneT& L, const float* __restrict__ p,
                                                 float fill, Slice<LaneT::NV>& s) {
#pragma unroll
  for (int v = 0; v < LaneT::NV; ++v)
#pragma unroll
    for (int i = 0; i < 4; ++i) s.a[v][i] = L.col_ok(v, i) ? p[L.col[v] + i] : fill;
}

// Native-RNG normals for this lane's slice at `step` (flat element e = chain*dim + col).
template <class LaneT>
__device__ __forceinline__ void normal_slice(const LaneT& L, RngKey key, uint64_t step,
                                             Slice<LaneT::NV>& s) {
  uint64_t row0 = (uint64_t)L.chain * (uint64_t)L.dim;
  // wide rows: hide the (loop-invariant) counter from the optimiser -- it otherwise hoists the counter words and the
  // first Philox multiply of all NV calls out of the caller's transition loop: 3 * NV registers that live through the
  // whole kernel, are spilled by the register-capped kernels and reloaded at every transition
  if constexpr (LaneT::NV >= 8) asm volatile("" : "+v"(row0));
#pragma unroll
  for (int v = 0; v < LaneT::NV; ++v) {
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
// this lane's slice.  Slots with !ok() hold x = 0 and must yield g = 0 and no energy.
// ---------------------------------------------------------------------------------
struct EnergyParams {
  int kind;
  int n_comp;
  int n_comp_pad;     // components staged in LDS (a multiple of 8: rows past n_comp are zero, their log-weight -inf)
  float s0, s1;
  const float* dev0;  // global
  const float* dev1;
  int param_in_lds;   // shared parameters were staged into LDS
  int dim_pad;        // row stride of the staged parameters (dim rounded up to 4)
  const int32_t* aux; // mixture: NULL or device int32[1], bit v = the component means differ somewhere in columns 4v..4v+3
};

// Internal energy kind (not part of the ABI): a mixture whose component means differ ONLY in the first four columns
// (bit mask == 1 in EnergyParams::aux) -- a K-mode mixture of a plane embedded in a higher-dimensional state, like the
// eight-mode ring of BASELINE config 3.  Columns all components share drop out of the responsibilities (softmax is
// shift-invariant) and their gradient is that of ONE Gaussian, so the two K x dim passes of the dense kernel shrink to
// K x 4: see Energy<kGmmSlot1>.  The one-lane-per-chain kernels look at the mask and pick the body themselves
// (a wave-uniform branch at the top of the kernel: no host read of device memory).
constexpr int kGmmSlot1 = 100;

__device__ __forceinline__ bool gmm_is_slot1(const EnergyParams& P) {
  if (P.aux == nullptr || P.n_comp > 8 || P.n_comp < 1) return false;
  return __builtin_amdgcn_readfirstlane(P.aux[0]) == 1;
}
// The one 4-column slot (0 .. 7) the component means differ in, or -1 (no hint, no or several slots): hmc_ring.hip runs the
// active-column kernels for ANY single slot (round 4; the shared body above only for slot 0).
__device__ __forceinline__ int gmm_single_slot(const EnergyParams& P) {
  if (P.aux == nullptr || P.n_comp > 8 || P.n_comp < 1) return -1;
  const int mask = __builtin_amdgcn_readfirstlane(P.aux[0]);
  if (mask <= 0 || mask > 0x80 || (mask & (mask - 1)) != 0) return -1;
  return __builtin_ctz((unsigned)mask);
}

// LDS carve-up (dynamic shared memory, 16-byte aligned):
//   [0, param_floats)                      shared parameters (P rows / mixture means + log-weights)
//   [param_floats, + waves * xchg_floats)  per-wave exchange rows (Gaussian only)
struct Smem {
  float* param;
  float* xchg;  // this wave's exchange buffer
};

template <int KIND, class LaneT>
struct Energy;

template <class LaneT>
struct Energy<EBM_ENERGY_DOUBLE_WELL, LaneT> {
  static constexpr int G = LaneT::G, NV = LaneT::NV;
  static constexpr bool HAS_GRAD_ONLY = false;
  float h, b2;
  __device__ __forceinline__ void init(const EnergyParams& P, const LaneT&, const Smem&) {
    h = P.s0; b2 = P.s1;
  }
  template <bool WANT_E>
  __device__ __forceinline__ float eval(const LaneT& L, const Slice<NV>& x, Slice<NV>& g) const {
    float acc = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xv = x.a[v][i];
        const float u = xv * xv - b2;
        const bool ok = L.ok(v, i);
        g.a[v][i] = ok ? ((4.0f * h) * u) * xv : 0.0f;  // == (h*(2u))*(2x) bit for bit (see langevin.hip: elem_grad)
        if (WANT_E) acc += ok ? u * u : 0.0f;
      }
    if (!WANT_E) return 0.0f;
    return h * group_sum<G>(acc);
  }
};

template <class LaneT>
struct Energy<EBM_ENERGY_HARMONIC, LaneT> {
  static constexpr int G = LaneT::G, NV = LaneT::NV;
  static constexpr bool HAS_GRAD_ONLY = false;
  float hk;
  __device__ __forceinline__ void init(const EnergyParams& P, const LaneT&, const Smem&) { hk = P.s0; }
  template <bool WANT_E>
  __device__ __forceinline__ float eval(const LaneT&, const Slice<NV>& x, Slice<NV>& g) const {
    float acc = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float xv = x.a[v][i];  // invalid slots are 0 and contribute 0
        g.a[v][i] = (2.0f * hk) * xv;  // == hk*(2x) bit for bit
        if (WANT_E) acc += xv * xv;
      }
    if (!WANT_E) return 0.0f;
    return hk * group_sum<G>(acc);
  }
};

// Gaussian: g = Ps d with Ps = (P + P^T)/2 (what autograd returns for 0.5 d^T P d), staged
// row-major in LDS; d is exchanged through the wave's LDS row so that every lane can walk
// all dim coordinates of its chain.  Ps symmetric => column slice of row j == needed block.
template <class LaneT>
struct Energy<EBM_ENERGY_GAUSSIAN, LaneT> {
  static constexpr int G = LaneT::G, NV = LaneT::NV;
  static constexpr bool HAS_GRAD_ONLY = false;
  Slice<NV> mu;
  const float* P_lds;
  const float* P_glb;
  float* xrow;  // this chain's exchange row in LDS
  int dim_pad;
  __device__ __forceinline__ void init(const EnergyParams& P, const LaneT& L, const Smem& S) {
    load_param_slice(L, P.dev0, 0.0f, mu);
    P_lds = P.param_in_lds ? S.param : nullptr;
    P_glb = P.dev1;
    dim_pad = P.dim_pad;
    xrow = S.xchg + L.chain_in_wave * (G * NV * 4);
    if constexpr (G == 1 && NV == 1) {  // dim <= 4: the whole precision matrix as 16 wave-uniform scalars
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int i = 0; i < 4; ++i) tiny[j][i] = (j < L.dim && i < L.dim) ? P.dev1[j * L.dim + i] : 0.0f;
    }
  }
  float tiny[(G == 1 && NV == 1) ? 4 : 1][4];  // Ps, zero-padded to 4x4 (dim <= 4 only)
  // dim <= 4, one lane per chain: no exchange row, no wave barrier, no LDS read -- 16 FMAs on scalars
  template <bool WANT_E>
  __device__ __forceinline__ float eval_tiny(const LaneT& L, const Slice<NV>& x, Slice<NV>& g) const {
    float d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      d[i] = L.ok(0, i) ? x.a[0][i] - mu.a[0][i] : 0.0f;
      g.a[0][i] = 0.0f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)   // same j-ascending FMA order as the LDS mat-vec
#pragma unroll
      for (int i = 0; i < 4; ++i) g.a[0][i] = __builtin_fmaf(tiny[j][i], d[j], g.a[0][i]);
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!L.ok(0, i)) g.a[0][i] = 0.0f;
      if (WANT_E) acc = __builtin_fmaf(d[i], g.a[0][i], acc);
    }
    return WANT_E ? 0.5f * acc : 0.0f;
  }
  template <bool WANT_E>
  __device__ __forceinline__ float eval(const LaneT& L, const Slice<NV>& x, Slice<NV>& g) const {
    if constexpr (G == 1 && NV == 1) return eval_tiny<WANT_E>(L, x, g);
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (P_lds) {
      auto row_fma = [&](int j, float dj) {
        const float* row = P_lds + j * dim_pad;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float4 pr = *reinterpret_cast<const float4*>(row + L.col[v]);
          g.a[v][0] = __builtin_fmaf(pr.x, dj, g.a[v][0]);
          g.a[v][1] = __builtin_fmaf(pr.y, dj, g.a[v][1]);
          g.a[v][2] = __builtin_fmaf(pr.z, dj, g.a[v][2]);
          g.a[v][3] = __builtin_fmaf(pr.w, dj, g.a[v][3]);
        }
      };
      // four rows per trip (one 16-byte read of d, four independent row reads in flight): the
      // one-row loop waited out a full LDS round trip per coordinate
      int j = 0;
#pragma unroll 2
      for (; j + 4 <= L.dim; j += 4) {
        const float4 d4 = *reinterpret_cast<const float4*>(xrow + j);
        row_fma(j, d4.x);
        row_fma(j + 1, d4.y);
        row_fma(j + 2, d4.z);
        row_fma(j + 3, d4.w);
      }
      for (; j < L.dim; ++j) row_fma(j, xrow[j]);
    } else {  // precision matrix too large for LDS: stream rows from L2
      for (int j = 0; j < L.dim; ++j) {
        const float dj = xrow[j];
        const float* row = P_glb + (int64_t)j * L.dim;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (L.col_ok(v, i)) g.a[v][i] = __builtin_fmaf(row[L.col[v] + i], dj, g.a[v][i]);
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
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

// Gaussian mixture (isotropic, shared sigma):  g = s1 * (x - sum_k r_k mu_k),  E = -(m + log sum_k e^{l_k - m}).
// Four evaluation paths, chosen by K, by where the parameters live and by the lane geometry:
//   K <= 8, staged, one lane per chain (G == 1, NV >= 4)  small_scalar_mu: means as scalar operands
//   K <= 8, staged, other geometries                      eval_small / grad_only: two branch-free passes
//   K > 10, staged                                        eval_blocks: blocks of eight, one rescale per block
//   otherwise (9-10 components, or parameters too large for LDS)  the per-component online-softmax loop
template <class LaneT>
struct Energy<EBM_ENERGY_GMM, LaneT> {
  static constexpr int G = LaneT::G, NV = LaneT::NV;
  static constexpr bool HAS_GRAD_ONLY = true;
  const float* mu_lds;
  const float* mu_glb;
  const float* logw;
  float lw8[8];
  float c8[8];  // lw8[k] - |mu_k|^2 / (2 sigma^2): the x-independent part of the logit (grad_only)
  int K, dim_pad;
  float inv2s2, invs2;
  __device__ __forceinline__ void init(const EnergyParams& P, const LaneT&, const Smem& S) {
    mu_lds = P.param_in_lds ? S.param : nullptr;
    mu_glb = P.dev0;
    logw = P.param_in_lds ? S.param + P.n_comp_pad * P.dim_pad : P.dev1;
    // small-mixture path: the (padded) log-weights are wave-uniform -> scalar loads, SGPRs
#pragma unroll
    for (int k = 0; k < 8; ++k) lw8[k] = (k < P.n_comp) ? P.dev1[k < P.n_comp ? k : 0] : -__builtin_inff();
    K = P.n_comp;
    dim_pad = P.dim_pad;
    inv2s2 = P.s0;
    invs2 = P.s1;
#pragma unroll
    for (int k = 0; k < 8; ++k) c8[k] = lw8[k];
    if (K <= 8 && mu_lds) {  // staged rows are zero-padded to 8 components x dim_pad columns
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float nrm = 0.0f;
        for (int d = 0; d < dim_pad; d += 4) {
          const float4 m = *reinterpret_cast<const float4*>(mu_lds + k * dim_pad + d);
          nrm = __builtin_fmaf(m.x, m.x, nrm);
          nrm = __builtin_fmaf(m.y, m.y, nrm);
          nrm = __builtin_fmaf(m.z, m.z, nrm);
          nrm = __builtin_fmaf(m.w, m.w, nrm);
        }
        // wave-uniform (every lane read the same LDS words): keep it on the scalar side -- as a vector
        // register it is spilled by the register-capped kernels and reloaded at every transition
        c8[k] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(__builtin_fmaf(-nrm, inv2s2, lw8[k]))));
      }
    }
  }
  __device__ __forceinline__ bool grad_only_ready() const { return K <= 8 && mu_lds != nullptr; }
  // Gradient without the energy (K <= 8, staged): softmax is shift-invariant, so |x|^2 drops out of
  // the logits and l_k = c8[k] + (x . mu_k) / sigma^2 -- one FMA per (component, coordinate) instead
  // of a subtract and an FMA, and a dot product is better conditioned than the difference form.
  // Returns sum_k exp(l_k - max): in [1, 8] for finite x, NaN as soon as one coordinate is not
  // (inf * 0, inf - inf or exp(inf - inf) appear on every route), and then g is unspecified.
  __device__ __forceinline__ float grad_only(const LaneT& L, const Slice<NV>& x, Slice<NV>& g) const {
    typedef float v2f __attribute__((ext_vector_type(2)));
    constexpr int KM = 8;
    if constexpr (G == 1 && LaneT::FULL && NV >= 4) return small_scalar_mu<false>(x, g);
    float logit[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      v2f dot = {0.0f, 0.0f};
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float4 m = *reinterpret_cast<const float4*>(mu_lds + k * dim_pad + L.col[v]);
        // slots beyond the row hold no state (ragged dims): keep them out of the sum
        const float x0 = L.ok(v, 0) ? x.a[v][0] : 0.0f, x1 = L.ok(v, 1) ? x.a[v][1] : 0.0f;
        const float x2 = L.ok(v, 2) ? x.a[v][2] : 0.0f, x3 = L.ok(v, 3) ? x.a[v][3] : 0.0f;
        // (their mu read lands in another row or in the -inf log-weight padding)
        const float m0 = L.ok(v, 0) ? m.x : 0.0f, m1 = L.ok(v, 1) ? m.y : 0.0f;
        const float m2 = L.ok(v, 2) ? m.z : 0.0f, m3 = L.ok(v, 3) ? m.w : 0.0f;
        dot = __builtin_elementwise_fma(v2f{x0, x1}, v2f{m0, m1}, dot);
        dot = __builtin_elementwise_fma(v2f{x2, x3}, v2f{m2, m3}, dot);
      }
      logit[k] = dot.x + dot.y;
    }
    float top = -__builtin_inff();
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      logit[k] = __builtin_fmaf(group_sum<G>(logit[k]), invs2, c8[k]);
      top = logit[k] > top ? logit[k] : top;
    }
    float sum = 0.0f;
    Slice<NV> acc;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc.a[v][i] = 0.0f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const float w = __expf(logit[k] - top);  // 0 for the padding components
      sum += w;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float4 m = *reinterpret_cast<const float4*>(mu_lds + k * dim_pad + L.col[v]);
        acc.a[v][0] = __builtin_fmaf(w, m.x, acc.a[v][0]);
        acc.a[v][1] = __builtin_fmaf(w, m.y, acc.a[v][1]);
        acc.a[v][2] = __builtin_fmaf(w, m.z, acc.a[v][2]);
        acc.a[v][3] = __builtin_fmaf(w, m.w, acc.a[v][3]);
      }
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        g.a[v][i] = L.ok(v, i) ? invs2 * (x.a[v][i] - acc.a[v][i] * inv) : 0.0f;
    return sum;
  }
  // One lane per chain (G == 1, full rows): every lane of the wave needs the SAME mu[k][d], so the
  // means are wave-uniform operands -- read through the scalar cache into SGPRs (constant address
  // space => s_load), no LDS traffic and no cross-lane reduction at all.
  // EXACT: logits in the reference's difference form -|x - mu_k|^2 / (2 sigma^2) and the energy as the
  // return value (H0 / H1 and diagnostics); otherwise the dot-product form, returning the softmax sum.
  template <bool EXACT>
  __device__ __forceinline__ float small_scalar_mu(const Slice<NV>& x, Slice<NV>& g) const {
    typedef float v2f __attribute__((ext_vector_type(2)));
    typedef float v16f __attribute__((ext_vector_type(16)));
    constexpr int KM = 8, D = NV * 4, CH = 16, PER_ROW = D / CH, NCH = KM * PER_ROW;
    static_assert(D % CH == 0, "rows stream in whole 16-float chunks");
    // The means stream through SGPRs in 16-float chunks (s_load_dwordx16), one chunk in flight while
    // the previous one is consumed.  The loads are volatile asm: left to itself the compiler hoists all
    // 2 x 8 x D scalar loads to the top of the block and spills 8*D SGPRs.
    const uint64_t base = (uint64_t)(uintptr_t)mu_glb;
    auto issue = [&](int c, v16f& dst) {
      const int k = c / PER_ROW, h = c % PER_ROW;
      const uint64_t src = base + (uint64_t)(((k < K ? k : K - 1) * D + h * CH) * 4);  // padding re-reads the last row
      asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(dst) : "s"(src));
    };
    auto arrive = [](v16f& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)); };
    float logit[KM];
    v16f buf[2];
    issue(0, buf[0]);
    v2f dot = {0.0f, 0.0f}, dot_b = {0.0f, 0.0f};  // two chains: a packed FMA cannot feed the next one back to back
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      arrive(buf[c & 1]);
      if (c + 1 < NCH) issue(c + 1, buf[(c + 1) & 1]);
      const v16f m = buf[c & 1];
      const int h = c % PER_ROW;
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const int v = h * (CH / 4) + q;
        if constexpr (EXACT) {
          const v2f da = v2f{x.a[v][0], x.a[v][1]} - v2f{m[4 * q], m[4 * q + 1]};
          const v2f db = v2f{x.a[v][2], x.a[v][3]} - v2f{m[4 * q + 2], m[4 * q + 3]};
          dot = __builtin_elementwise_fma(da, da, dot);
          dot_b = __builtin_elementwise_fma(db, db, dot_b);
        } else {
          dot = __builtin_elementwise_fma(v2f{x.a[v][0], x.a[v][1]}, v2f{m[4 * q], m[4 * q + 1]}, dot);
          dot_b = __builtin_elementwise_fma(v2f{x.a[v][2], x.a[v][3]}, v2f{m[4 * q + 2], m[4 * q + 3]}, dot_b);
        }
      }
      if (h == PER_ROW - 1) {
        dot += dot_b;
        if constexpr (EXACT) logit[c / PER_ROW] = __builtin_fmaf(-(dot.x + dot.y), inv2s2, lw8[c / PER_ROW]);
        else logit[c / PER_ROW] = __builtin_fmaf(dot.x + dot.y, invs2, c8[c / PER_ROW]);
        dot = dot_b = v2f{0.0f, 0.0f};
      }
      asm volatile("" : "+v"(dot), "+v"(dot_b));  // the chunk's FMAs stay between its load and the next one
      __builtin_amdgcn_sched_barrier(0);
    }
    float top = logit[0];
#pragma unroll
    for (int k = 1; k < KM; ++k) top = __builtin_fmaxf(top, logit[k]);  // a NaN logit resurfaces in the sum
    float w[KM];
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      w[k] = __expf(logit[k] - top);
      sum += w[k];
    }
    v2f acc[NV][2];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v][0] = acc[v][1] = v2f{0.0f, 0.0f};
    issue(0, buf[0]);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      arrive(buf[c & 1]);
      if (c + 1 < NCH) issue(c + 1, buf[(c + 1) & 1]);
      const v16f m = buf[c & 1];
      const int h = c % PER_ROW;
      const v2f wk = {w[c / PER_ROW], w[c / PER_ROW]};
#pragma unroll
      for (int q = 0; q < CH / 4; ++q) {
        const int v = h * (CH / 4) + q;
        acc[v][0] = __builtin_elementwise_fma(wk, v2f{m[4 * q], m[4 * q + 1]}, acc[v][0]);
        acc[v][1] = __builtin_elementwise_fma(wk, v2f{m[4 * q + 2], m[4 * q + 3]}, acc[v][1]);
        asm volatile("" : "+v"(acc[v][0]), "+v"(acc[v][1]));  // keep the FMAs with their chunk
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const float inv = EXACT ? 1.0f / sum : __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      g.a[v][0] = invs2 * (x.a[v][0] - acc[v][0].x * inv);
      g.a[v][1] = invs2 * (x.a[v][1] - acc[v][0].y * inv);
      g.a[v][2] = invs2 * (x.a[v][2] - acc[v][1].x * inv);
      g.a[v][3] = invs2 * (x.a[v][3] - acc[v][1].y * inv);
    }
    if constexpr (EXACT) return -(top + logf(sum));
    return sum;
  }
  __device__ __forceinline__ void load_mu(const LaneT& L, int k, int v, float (&m)[4]) const {
    if (mu_lds) {
      const float4 t = *reinterpret_cast<const float4*>(mu_lds + k * dim_pad + L.col[v]);
      m[0] = t.x; m[1] = t.y; m[2] = t.z; m[3] = t.w;
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        m[i] = L.col_ok(v, i) ? mu_glb[(int64_t)k * L.dim + L.col[v] + i] : 0.0f;
    }
  }
  // K <= 8 (staged in LDS, padded to exactly 8 components whose log-weight is -inf): two
  // branch-free passes over register-held logits.  All 8 distances and their cross-lane
  // reductions are independent, so they pipeline; no running-max dependency chain and one
  // exp per component instead of two.
  template <bool WANT_E>
  __device__ __forceinline__ float eval_small(const LaneT& L, const Slice<NV>& x, Slice<NV>& g) const {
    constexpr int KM = 8;
    if constexpr (G == 1 && LaneT::FULL && NV >= 4) return small_scalar_mu<true>(x, g);
    float logit[KM];
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      float dist = 0.0f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float4 m = *reinterpret_cast<const float4*>(mu_lds + k * dim_pad + L.col[v]);
        const float mk[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float df = L.ok(v, i) ? x.a[v][i] - mk[i] : 0.0f;
          dist = __builtin_fmaf(df, df, dist);
        }
      }
      logit[k] = dist;
    }
    float top = -__builtin_inff();
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      logit[k] = __builtin_fmaf(-group_sum<G>(logit[k]), inv2s2, lw8[k]);
      top = logit[k] > top ? logit[k] : top;
    }
    float sum = 0.0f;
    Slice<NV> acc;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc.a[v][i] = 0.0f;
#pragma unroll
    for (int k = 0; k < KM; ++k) {
      const float w = __expf(logit[k] - top);  // 0 for the padding components
      sum += w;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float4 m = *reinterpret_cast<const float4*>(mu_lds + k * dim_pad + L.col[v]);
        acc.a[v][0] = __builtin_fmaf(w, m.x, acc.a[v][0]);
        acc.a[v][1] = __builtin_fmaf(w, m.y, acc.a[v][1]);
        acc.a[v][2] = __builtin_fmaf(w, m.z, acc.a[v][2]);
        acc.a[v][3] = __builtin_fmaf(w, m.w, acc.a[v][3]);
      }
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i)
        g.a[v][i] = L.ok(v, i) ? invs2 * (x.a[v][i] - acc.a[v][i] * inv) : 0.0f;
    if (!WANT_E) return 0.0f;
    return -(top + logf(sum));
  }

  // K > 8, staged: the components in blocks of eight.  Inside a block the eight distances, their
  // cross-lane reductions and exps are independent (as in eval_small); blocks are chained by ONE running
  // max / rescale per block instead of one per component (the generic loop below pays a dependent
  // exp + rescale of the whole accumulator for every component: 4x slower per component).
  // Rows are staged zero-padded to a multiple of eight with log-weight -inf.
  template <bool WANT_E>
  __device__ __forceinline__ float eval_blocks(const LaneT& L, const Slice<NV>& x, Slice<NV>& g) const {
    constexpr int KB = 8;
    const int n_blocks = (K + KB - 1) / KB;
    float run_max = -__builtin_inff();
    float run_sum = 0.0f;
    Slice<NV> acc;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc.a[v][i] = 0.0f;
    for (int b = 0; b < n_blocks; ++b) {
      const float* rows = mu_lds + (b * KB) * dim_pad;
      float logit[KB];
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        float dist = 0.0f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float4 m = *reinterpret_cast<const float4*>(rows + k * dim_pad + L.col[v]);
          const float mk[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float df = L.ok(v, i) ? x.a[v][i] - mk[i] : 0.0f;
            dist = __builtin_fmaf(df, df, dist);
          }
        }
        logit[k] = dist;
      }
      float top = run_max;
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        logit[k] = __builtin_fmaf(-group_sum<G>(logit[k]), inv2s2, logw[b * KB + k]);
        top = logit[k] > top ? logit[k] : top;
      }
      const float scale = __expf(run_max - top);  // 0 on the first block
      run_sum *= scale;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc.a[v][i] *= scale;
#pragma unroll
      for (int k = 0; k < KB; ++k) {
        const float w = __expf(logit[k] - top);  // 0 for the padding components
        run_sum += w;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
          const float4 m = *reinterpret_cast<const float4*>(rows + k * dim_pad + L.col[v]);
          acc.a[v][0] = __builtin_fmaf(w, m.x, acc.a[v][0]);
          acc.a[v][1] = __builtin_fmaf(w, m.y, acc.a[v][1]);
          acc.a[v][2] = __builtin_fmaf(w, m.z, acc.a[v][2]);
          acc.a[v][3] = __builtin_fmaf(w, m.w, acc.a[v][3]);
        }
      }
      run_max = top;
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

  template <bool WANT_E>
  __device__ __forceinline__ float eval(const LaneT& L, const Slice<NV>& x, Slice<NV>& g) const {
    // (not compiled into the one-lane-per-chain kernels: the launchers only pick that geometry for K <= 8,
    //  and the extra code cost the config-3 kernel 12 % through register pressure)
    if constexpr (!(G == 1 && LaneT::FULL && NV >= 4)) {
      if (K > 10 && mu_lds) return eval_blocks<WANT_E>(L, x, g);  // 9 or 10 components: two padded blocks cost more than the loop below
    }
    if (K <= 8 && mu_lds) {
      if constexpr (!WANT_E) {
        grad_only(L, x, g);
        return 0.0f;
      } else {
        return eval_small<true>(L, x, g);
      }
    }
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

// Mixture with the component means differing in columns 0..3 only (kGmmSlot1; one lane per chain, full rows).
//   logits  l_k = c_k + (x[0:4] . mu_k[0:4]) / sigma^2        (the |x|^2 term and every shared column cancel in softmax)
//   gradient    g[0:4] = (x[0:4] - sum_k r_k mu_k[0:4]) / sigma^2,   g[d] = (x[d] - mu_0[d]) / sigma^2 for d >= 4
//   energy      E = sum_{d >= 4} (x[d] - mu_0[d])^2 / (2 sigma^2)  -  logsumexp_k(logw_k - |x[0:4] - mu_k[0:4]|^2 / (2 sigma^2))
// The 8 x 4 active means stay RESIDENT in 32 SGPRs for the whole kernel (no load in the step loop); the shared
// row mu_0 streams through 16 SGPRs per chunk, one wait per 16 columns.  ~140 VALU instructions per gradient
// where the dense passes take ~380 plus 32 scalar-load round trips.
template <class LaneT>
struct Energy<kGmmSlot1, LaneT> {
  static constexpr int G = LaneT::G, NV = LaneT::NV;
  static constexpr bool HAS_GRAD_ONLY = true;
  static_assert(G == 1 && LaneT::FULL && NV >= 4 && NV % 4 == 0, "one lane per chain, rows in whole 16-float chunks");
  typedef float v2f __attribute__((ext_vector_type(2)));
  typedef float v16f __attribute__((ext_vector_type(16)));
  float mact[8][4];  // wave-uniform: scalar registers
  float lw8[8], c8[8];
  const float* mu_glb;
  float inv2s2, invs2;
  __device__ __forceinline__ void init(const EnergyParams& P, const LaneT&, const Smem&) {
    constexpr int D = NV * 4;
    mu_glb = P.dev0;
    inv2s2 = P.s0;
    invs2 = P.s1;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const int kk = k < P.n_comp ? k : P.n_comp - 1;  // padding re-reads the last row, its log-weight is -inf
      lw8[k] = (k < P.n_comp) ? P.dev1[kk] : -__builtin_inff();
      float nrm = 0.0f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        mact[k][i] = P.dev0[kk * D + i];
        nrm = __builtin_fmaf(mact[k][i], mact[k][i], nrm);
      }
      c8[k] = __builtin_fmaf(-nrm, inv2s2, lw8[k]);
    }
  }
  __device__ __forceinline__ bool grad_only_ready() const { return true; }
  // One evaluation.  Row 0 of the means (the shared columns) is requested FIRST, the active slot is worked out
  // while the scalar loads are in flight (~100 instructions), then the shared columns: g = (x - mu_0) / sigma^2 and,
  // with the energy, sum (x - mu_0)^2.  The gradient-only form returns the softmax sum of the active slot: finite
  // => the active coordinates are finite.  A shared coordinate cannot turn NaN inside a trajectory (x + eps p with
  // finite p), only +-inf, and that is absorbing: it surfaces in the energy of the trajectory's last evaluation,
  // whose literal path scrubs it -- the proposal is rejected either way (H1 at its clamp).
  template <bool EXACT>
  __device__ __forceinline__ float evaluate(const Slice<NV>& x, Slice<NV>& g) const {
    constexpr int NCH = NV / 4;  // 16-float chunks of row 0
    const uint64_t base = (uint64_t)(uintptr_t)mu_glb;
    v16f row0[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(row0[c]) : "s"(base + (uint64_t)(c * 64)));

    const v2f xa = {x.a[0][0], x.a[0][1]}, xb = {x.a[0][2], x.a[0][3]};
    float logit[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const v2f ma = {mact[k][0], mact[k][1]}, mb = {mact[k][2], mact[k][3]};
      if constexpr (EXACT) {
        const v2f da = xa - ma, db = xb - mb;
        const v2f d2 = __builtin_elementwise_fma(db, db, da * da);
        logit[k] = __builtin_fmaf(-(d2.x + d2.y), inv2s2, lw8[k]);
      } else {
        const v2f dt = __builtin_elementwise_fma(xb, mb, xa * ma);
        logit[k] = __builtin_fmaf(dt.x + dt.y, invs2, c8[k]);
      }
    }
    float top = logit[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) top = __builtin_fmaxf(top, logit[k]);  // a NaN logit resurfaces in the sum
    float w[8], sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      w[k] = __expf(logit[k] - top);
      sum += w[k];
    }
    v2f acc_a = {0.0f, 0.0f}, acc_b = {0.0f, 0.0f};
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const v2f wk = {w[k], w[k]};
      acc_a = __builtin_elementwise_fma(wk, v2f{mact[k][0], mact[k][1]}, acc_a);
      acc_b = __builtin_elementwise_fma(wk, v2f{mact[k][2], mact[k][3]}, acc_b);
    }
    const float inv = EXACT ? 1.0f / sum : __builtin_amdgcn_rcpf(sum);
    g.a[0][0] = invs2 * (x.a[0][0] - acc_a.x * inv);
    g.a[0][1] = invs2 * (x.a[0][1] - acc_a.y * inv);
    g.a[0][2] = invs2 * (x.a[0][2] - acc_b.x * inv);
    g.a[0][3] = invs2 * (x.a[0][3] - acc_b.y * inv);

    // the shared columns
#pragma unroll
    for (int c = 0; c < NCH; ++c) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(row0[c]));
    v2f sq = {0.0f, 0.0f}, sq_b = {0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int v = c * 4 + q;
        if (v == 0) continue;  // the active slot
        const v2f da = v2f{x.a[v][0], x.a[v][1]} - v2f{row0[c][4 * q], row0[c][4 * q + 1]};
        const v2f db = v2f{x.a[v][2], x.a[v][3]} - v2f{row0[c][4 * q + 2], row0[c][4 * q + 3]};
        if constexpr (EXACT) {
          sq = __builtin_elementwise_fma(da, da, sq);
          sq_b = __builtin_elementwise_fma(db, db, sq_b);
        }
        g.a[v][0] = invs2 * da.x; g.a[v][1] = invs2 * da.y;
        g.a[v][2] = invs2 * db.x; g.a[v][3] = invs2 * db.y;
      }
    }
    sq += sq_b;
    const float common_sq = sq.x + sq.y;
    if constexpr (EXACT) return __builtin_fmaf(common_sq, inv2s2, -(top + logf(sum)));
    return sum;
  }
  __device__ __forceinline__ float grad_only(const LaneT&, const Slice<NV>& x, Slice<NV>& g) const { return evaluate<false>(x, g); }
  template <bool WANT_E>
  __device__ __forceinline__ float eval(const LaneT&, const Slice<NV>& x, Slice<NV>& g) const {
    if constexpr (!WANT_E) {
      evaluate<false>(x, g);
      return 0.0f;
    } else {
      return evaluate<true>(x, g);
    }
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
    const int n = P.n_comp_pad * P.dim_pad;
    for (int i = threadIdx.x; i < n; i += kBlock) {
      const int r = i / P.dim_pad, c = i - r * P.dim_pad;
      dst[i] = (c < dim && r < P.n_comp) ? P.dev0[(int64_t)r * dim + c] : 0.0f;
    }
    for (int i = threadIdx.x; i < P.n_comp_pad; i += kBlock)
      dst[n + i] = (i < P.n_comp) ? P.dev1[i] : -__builtin_inff();
  }
  __syncthreads();
}

template <int NV>
__device__ __forceinline__ Smem carve_smem(float* smem, int param_floats) {
  Smem S;
  S.param = smem;
  S.xchg = smem + param_floats + (threadIdx.x >> 6) * (64 * NV * 4);
  return S;
}

// ---------------------------------------------------------------------------------
// host side: geometry selection and launch planning
// ---------------------------------------------------------------------------------
struct Geometry {
  int G, NV;
  bool full;
};

inline bool pick_geometry(int dim, Geometry& geo) {
  const int nvec = (dim + 3) / 4;
  if (nvec <= 64) {
    int g = 1;
    while (g < nvec) g <<= 1;
    geo = Geometry{g, 1, false};
  } else if (nvec <= 128) {
    geo = Geometry{64, 2, false};
  } else if (nvec <= 256) {
    geo = Geometry{64, 4, false};
  } else {
    return false;
  }
  geo.full = dim == 4 * geo.G * geo.NV;
  return true;
}

// Decide where the shared parameters live and how much dynamic LDS the launch needs.
inline void plan_params(const ebm_energy_t& e, int dim, const Geometry& geo, EnergyParams& P,
                        int& param_floats, size_t& smem_bytes) {
  P.kind = e.kind; P.n_comp = e.n_comp; P.s0 = e.s[0]; P.s1 = e.s[1];
  P.n_comp_pad = e.n_comp < 8 ? 8 : ((e.n_comp + 7) & ~7);  // whole blocks of eight (see Energy<GMM>::eval_blocks)
  P.dev0 = e.dev0; P.dev1 = e.dev1; P.aux = e.aux;
  P.dim_pad = (dim + 3) & ~3;
  P.param_in_lds = 0;
  param_floats = 0;
  size_t xchg = 0;
  if (e.kind == EBM_ENERGY_GAUSSIAN) {
    const size_t need = (size_t)dim * P.dim_pad;
    if (need * 4 <= (size_t)kParamLdsBudget) { P.param_in_lds = 1; param_floats = (int)need; }
    xchg = (size_t)kWavesPerBlock * 64 * geo.NV * 4;
  } else if (e.kind == EBM_ENERGY_GMM) {
    const size_t need = (size_t)P.n_comp_pad * P.dim_pad + (size_t)((P.n_comp_pad + 3) & ~3);
    if (need * 4 <= (size_t)kParamLdsBudget) { P.param_in_lds = 1; param_floats = (int)need; }
  }
  smem_bytes = ((size_t)param_floats + xchg) * sizeof(float);
}

inline int64_t blocks_for(int64_t n_chains, const Geometry& geo) {
  return ceil_div64(n_chains, kBlock / geo.G);
}

// KERNEL<KIND, G, NV, FULL> dispatch.  FULL variants exist for NV == 1 only (dim <= 256);
// larger rows always carry their validity mask.
#define EBM_GEO_LAUNCH(KERNEL, KIND, geo, ...)                                                   \
  do {                                                                                           \
    if (geo.NV == 1 && geo.full) {                                                               \
      switch (geo.G) {                                                                           \
        case 1:  hipLaunchKernelGGL((KERNEL<KIND, 1, 1, true>), __VA_ARGS__); break;             \
        case 2:  hipLaunchKernelGGL((KERNEL<KIND, 2, 1, true>), __VA_ARGS__); break;             \
        case 4:  hipLaunchKernelGGL((KERNEL<KIND, 4, 1, true>), __VA_ARGS__); break;             \
        case 8:  hipLaunchKernelGGL((KERNEL<KIND, 8, 1, true>), __VA_ARGS__); break;             \
        case 16: hipLaunchKernelGGL((KERNEL<KIND, 16, 1, true>), __VA_ARGS__); break;            \
        case 32: hipLaunchKernelGGL((KERNEL<KIND, 32, 1, true>), __VA_ARGS__); break;            \
        default: hipLaunchKernelGGL((KERNEL<KIND, 64, 1, true>), __VA_ARGS__); break;            \
      }                                                                                          \
    } else if (geo.NV == 1) {                                                                    \
      switch (geo.G) {                                                                           \
        case 1:  hipLaunchKernelGGL((KERNEL<KIND, 1, 1, false>), __VA_ARGS__); break;            \
        case 2:  hipLaunchKernelGGL((KERNEL<KIND, 2, 1, false>), __VA_ARGS__); break;            \
        case 4:  hipLaunchKernelGGL((KERNEL<KIND, 4, 1, false>), __VA_ARGS__); break;            \
        case 8:  hipLaunchKernelGGL((KERNEL<KIND, 8, 1, false>), __VA_ARGS__); break;            \
        case 16: hipLaunchKernelGGL((KERNEL<KIND, 16, 1, false>), __VA_ARGS__); break;           \
        case 32: hipLaunchKernelGGL((KERNEL<KIND, 32, 1, false>), __VA_ARGS__); break;           \
        default: hipLaunchKernelGGL((KERNEL<KIND, 64, 1, false>), __VA_ARGS__); break;           \
      }                                                                                          \
    } else if (geo.NV == 2) {                                                                    \
      hipLaunchKernelGGL((KERNEL<KIND, 64, 2, false>), __VA_ARGS__);                             \
    } else {                                                                                     \
      hipLaunchKernelGGL((KERNEL<KIND, 64, 4, false>), __VA_ARGS__);                             \
    }                                                                                            \
  } while (0)

#define EBM_KIND_LAUNCH(KERNEL, kind, geo, ...)                                                  \
  do {                                                                                           \
    switch (kind) {                                                                              \
      case EBM_ENERGY_DOUBLE_WELL: EBM_GEO_LAUNCH(KERNEL, EBM_ENERGY_DOUBLE_WELL, geo, __VA_ARGS__); break; \
      case EBM_ENERGY_HARMONIC:    EBM_GEO_LAUNCH(KERNEL, EBM_ENERGY_HARMONIC, geo, __VA_ARGS__); break;    \
      case EBM_ENERGY_GAUSSIAN:    EBM_GEO_LAUNCH(KERNEL, EBM_ENERGY_GAUSSIAN, geo, __VA_ARGS__); break;    \
      default:                     EBM_GEO_LAUNCH(KERNEL, EBM_ENERGY_GMM, geo, __VA_ARGS__); break;         \
    }                                                                                            \
  } while (0)

}  // namespace rows
}  // namespace ebm
