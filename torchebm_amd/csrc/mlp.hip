// k-fused Langevin chain for a two-hidden-layer SiLU MLP energy:
//     E(x) = w3 . silu(W2 silu(W1 x + b1) + b2) + b3,      x in R^dim (dim <= 4), hidden width H = 128
// forward AND input-gradient inside the kernel, weights resident in LDS for all k steps, the two
// H x H contractions on the matrix cores with the exact-f32 MFMA (v_mfma_f32_32x32x2_f32), so the
// sampler no longer round-trips through autograd (~30 launches) every Langevin step.
// Reference shape: examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-31 (the energy),
// torchebm/samplers/langevin_dynamics.py:154-185 (the loop), core/base_integrator.py:711-731 (the update).
//
// Mapping.  A wavefront owns 32 chains ("samples"); lane l = (m, h) with m = l & 31 the sample and
// h = l >> 5 the K-half of the 32x32x2 MFMA.  Everything is computed TRANSPOSED so that the sample
// index stays on the lane axis through both GEMMs and no layout change is ever needed:
//   forward   A2^T[j, m] = sum_i W2[j, i] * h1[i, m]      A-operand = W2 (LDS), B-operand = h1 computed on the
//                                                         fly from x (2 FMAs + SiLU per value, reused by 4 tiles)
//   backward  T^T[i, m]  = sum_j W2[j, i] * d2[j, m]      B-operand = d2 = w3 * silu'(a2), read straight out of
//                                                         the forward accumulators: the K index j is enumerated
//                                                         in the order the C/D layout already holds it
//   g[m, c]   = sum_i W1[i, c] * T^T[i, m] * silu'(a1[i, m])   VALU + one cross-half add
// C/D layout of the 32x32 tile: register r of lane (m, h) holds row (r&3) + 8*(r>>2) + 4*h, column m.
// LDS: W2 with row stride H+1 (conflict-free for both the row-walk of the forward A-operand and the
// column-walk of the backward one), W1 rows padded to 4 floats with b1 appended, b2, w3.
// A block = 4 waves = 128 chains, 69.6 KiB of LDS -> two blocks per CU.
#include "ebm_common.h"

namespace ebm {

bool mlp_wide_supported(int32_t hidden, int32_t dim);
bool mlp_wide_hmc_supported(int32_t hidden, int32_t dim);  // mlp_wide_hmc.hip
int launch_hmc_chain_mlp_wide(int32_t hidden, const float* params, float* x, int64_t n_chains, int32_t dim, int32_t n_mh,
                              int32_t n_leapfrog, float eps, const float* eps_table, int32_t mass_kind, double mass_scalar,
                              const float* mass_diag, int32_t thin, float* traj, uint8_t* accept_mask, uint32_t* accept_count,
                              const float* p_noise, const float* u, uint64_t seed, uint64_t offset, hipStream_t st, const char* who);
int launch_mlp_wide(int32_t hidden, const float* params, float* x, int64_t n_chains, int32_t dim, int32_t k_steps, float eta,
                    float sqrt_eta, float noise_coef, const float* coef_table, int clamp_on, float cmin, float cmax,
                    int32_t thin, float* traj, const float* noise, uint64_t seed, uint64_t offset, float* energy_out,
                    float* grad_out, hipStream_t st, const char* who);

namespace {

constexpr int kBlock = 256;
constexpr int H = 128;            // hidden width (both layers)
constexpr int kTiles = H / 32;    // 32-row tiles of a hidden vector
constexpr int kW2Stride = H + 1;
constexpr int kMaxDim = 4;

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct MlpArgs {
  float* x;              // [n, dim] in/out (k_steps > 0) or input (k_steps == 0)
  int64_t n_chains;
  int32_t dim;
  int32_t k_steps;
  float eta, sqrt_eta, noise_coef;
  const float4* table;
  int clamp_on;
  float cmin, cmax;
  int32_t thin, n_kept;
  float* traj;
  const float* noise;    // [k, n, dim] or null
  RngKey key;
  uint64_t step0;
  const float* params;   // packed W1[H,dim] b1[H] W2[H,H] b2[H] w3[H] b3[1]
  float* energy_out;     // k_steps == 0: E(x)[n]
  float* grad_out;       // k_steps == 0: dE/dx[n, dim]
};

extern __shared__ __attribute__((aligned(16))) float mlp_smem[];

__device__ __forceinline__ float sigmoidf_fast(float a) { return __builtin_amdgcn_rcpf(1.0f + __expf(-a)); }

__device__ __forceinline__ int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(kBlock, 2) void mlp_langevin_chain_kernel(MlpArgs a) {
  float* W2s = mlp_smem;                       // [H][H+1]
  float* W1s = W2s + H * kW2Stride;            // [H][8]: W1 row (padded to 4), b1, 3 unused
  float* b2s = W1s + H * 8;                    // [H]
  float* w3s = b2s + H;                        // [H]
  const int dim = a.dim;
  {  // stage the weights (once per launch)
    const float* W1g = a.params;
    const float* b1g = W1g + H * dim;
    const float* W2g = b1g + H;
    const float* b2g = W2g + H * H;
    const float* w3g = b2g + H;
    for (int i = threadIdx.x; i < H * H; i += kBlock) W2s[(i / H) * kW2Stride + (i % H)] = W2g[i];
    for (int i = threadIdx.x; i < H; i += kBlock) {
#pragma unroll
      for (int c = 0; c < kMaxDim; ++c) W1s[i * 8 + c] = c < dim ? W1g[i * dim + c] : 0.0f;
      W1s[i * 8 + 4] = b1g[i];
      b2s[i] = b2g[i];
      w3s[i] = w3g[i];
    }
    __syncthreads();
  }
  const float b3 = a.params[H * dim + H + H * H + H + H];

  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int64_t sample = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 32 + m;
  const bool active = sample < a.n_chains;

  float x[kMaxDim];
#pragma unroll
  for (int c = 0; c < kMaxDim; ++c) x[c] = (active && c < dim) ? a.x[sample * dim + c] : 0.0f;

  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  const int n_evals = a.k_steps > 0 ? a.k_steps : 1;

  for (int step = 0; step < n_evals; ++step) {
#include "mlp_eval_body.inc"

    if (a.k_steps == 0) {  // evaluation only
      if (active && h == 0) {
        if (a.energy_out) a.energy_out[sample] = energy;
        if (a.grad_out)
          for (int c = 0; c < dim; ++c) a.grad_out[sample * dim + c] = g[c];
      }
      return;
    }

    // ------------------------------------------------------------ Euler-Maruyama update (reference op order)
    if (a.table) {
      const float4 tb = a.table[step];
      eta = tb.x; sqrt_eta = tb.y; noise_coef = tb.z;
    }
    float eps[kMaxDim] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.noise) {
      if (active)
        for (int c = 0; c < dim; ++c) eps[c] = a.noise[((int64_t)step * a.n_chains + sample) * dim + c];
    } else {
      uint64_t have = ~0ull;
      F4 nrm;
      for (int c = 0; c < dim; ++c) {
        const uint64_t e = (uint64_t)sample * (uint64_t)dim + (uint64_t)c;
        if ((e >> 2) != have) {
          have = e >> 2;
          nrm = normal4_at(a.key, have, a.step0 + (uint64_t)step);
        }
        const int q = (int)(e & 3);
        eps[c] = q == 0 ? nrm.v[0] : (q == 1 ? nrm.v[1] : (q == 2 ? nrm.v[2] : nrm.v[3]));
      }
    }
#pragma unroll
    for (int c = 0; c < kMaxDim; ++c) {
      const float x1 = x[c] - eta * g[c];
      const float dw = eps[c] * sqrt_eta;
      float nv = x1 + noise_coef * dw;
      if (a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
      x[c] = (c < dim) ? nv : 0.0f;
    }
    if (a.traj && --until_keep == 0) {
      until_keep = a.thin;
      if (active && h == 0)
        for (int c = 0; c < dim; ++c) a.traj[sample * (int64_t)a.n_kept * dim + keep_off + c] = x[c];
      keep_off += dim;
    }
  }
  if (active && h == 0)
    for (int c = 0; c < dim; ++c) a.x[sample * dim + c] = x[c];
}

// ---------------------------------------------------------------------------------
// HMC transitions on the same energy (samplers/hmc.py:201-315 + integrators/leapfrog.py:116-187): the
// evaluation block above sits ONCE in a small state machine -- mode 0: E and force at the current state,
// mode 1: after a kick + drift, mode 2: re-evaluation on a scrubbed position (safe mode's literal path) --
// so that every MFMA is reached by the whole wave whatever single chains do.  dim <= 4: a lane holds a
// whole chain (x, p, f in 12 registers), both K-halves of a sample carry identical copies.
// RNG coordinates as in hmc_kernel.h: momentum at step 2t, uniforms at 2t+1.
// ---------------------------------------------------------------------------------
struct MlpHmcArgs {
  float* x;
  int64_t n_chains;
  int32_t dim, n_mh, n_leapfrog;
  float eps;
  const float* eps_table;
  int32_t mass_kind;
  float mass_raw, mass_sqrt, mass_safe;
  const float* mass_diag;
  int32_t thin, n_kept;
  float* traj;
  uint8_t* accept_mask;
  uint32_t* accept_count;
  const float* p_noise;
  const float* u;
  RngKey key;
  uint64_t step0;
  const float* params;
};

__global__ __launch_bounds__(kBlock, 2) void mlp_hmc_chain_kernel(MlpHmcArgs a) {
  float* W2s = mlp_smem;
  float* W1s = W2s + H * kW2Stride;
  float* b2s = W1s + H * 8;
  float* w3s = b2s + H;
  const int dim = a.dim;
  {  // stage the weights (once per launch)
    const float* W1g = a.params;
    const float* b1g = W1g + H * dim;
    const float* W2g = b1g + H;
    const float* b2g = W2g + H * H;
    const float* w3g = b2g + H;
    for (int i = threadIdx.x; i < H * H; i += kBlock) W2s[(i / H) * kW2Stride + (i % H)] = W2g[i];
    for (int i = threadIdx.x; i < H; i += kBlock) {
#pragma unroll
      for (int c = 0; c < kMaxDim; ++c) W1s[i * 8 + c] = c < dim ? W1g[i * dim + c] : 0.0f;
      W1s[i * 8 + 4] = b1g[i];
      b2s[i] = b2g[i];
      w3s[i] = w3g[i];
    }
    __syncthreads();
  }
  const float b3 = a.params[H * dim + H + H * H + H + H];

  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int64_t sample = ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 32 + m;
  const bool active = sample < a.n_chains;

  float xc[kMaxDim], m_raw[kMaxDim], m_sqrt[kMaxDim], m_safe[kMaxDim];
#pragma unroll
  for (int c = 0; c < kMaxDim; ++c) {
    xc[c] = (active && c < dim) ? a.x[sample * dim + c] : 0.0f;
    float mr = 1.0f;
    if (a.mass_kind == EBM_MASS_SCALAR) mr = a.mass_raw;
    else if (a.mass_kind == EBM_MASS_DIAG) mr = c < dim ? a.mass_diag[c] : 1.0f;
    m_raw[c] = mr;
    m_sqrt[c] = a.mass_kind == EBM_MASS_SCALAR ? a.mass_sqrt : sqrtf(mr);
    m_safe[c] = a.mass_kind == EBM_MASS_SCALAR ? a.mass_safe : (mr < 1e-10f ? 1e-10f : mr);
  }
  const bool has_mass = a.mass_kind != EBM_MASS_NONE;
  const bool diag_mass = a.mass_kind == EBM_MASS_DIAG;

  auto kinetic = [&](const float (&q)[kMaxDim]) -> float {  // 0.5 p^T M^-1 p, clamped to [0, 1e10]
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < kMaxDim; ++c) {
      float sq = q[c] * q[c];
      if (diag_mass) sq = sq / m_raw[c];
      acc += sq;
    }
    float k = 0.5f * acc;
    if (has_mass && !diag_mass) k = k / a.mass_raw;
    return clamp_nanprop(k, 0.0f, 1e10f);
  };

  int until_keep = a.thin;
  int64_t keep_off = 0;
  float eps = a.eps;

  for (int tr = 0; tr < a.n_mh; ++tr) {
    if (a.eps_table) eps = a.eps_table[tr];
    const float half_eps = 0.5f * eps;

    // ---- momentum draw p ~ N(0, M)
    float p[kMaxDim] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (a.p_noise) {
      if (active)
        for (int c = 0; c < dim; ++c) p[c] = a.p_noise[((int64_t)tr * a.n_chains + sample) * dim + c];
    } else {
      uint64_t have = ~0ull;
      F4 nrm;
      for (int c = 0; c < dim; ++c) {
        const uint64_t e = (uint64_t)sample * (uint64_t)dim + (uint64_t)c;
        if ((e >> 2) != have) {
          have = e >> 2;
          nrm = normal4_at(a.key, have, a.step0 + 2ull * (uint64_t)tr);
        }
        const int q = (int)(e & 3);
        p[c] = q == 0 ? nrm.v[0] : (q == 1 ? nrm.v[1] : (q == 2 ? nrm.v[2] : nrm.v[3]));
      }
    }
    if (has_mass) {
#pragma unroll
      for (int c = 0; c < kMaxDim; ++c) p[c] *= m_sqrt[c];
    }

    float x[kMaxDim], f[kMaxDim] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int c = 0; c < kMaxDim; ++c) x[c] = xc[c];
    float h0 = 0.0f, e_last = 0.0f;
    int done = 0, mode = 0;  // wave-uniform
    while (done <= a.n_leapfrog) {
      if (mode == 1) {  // first half kick + drift
#pragma unroll
        for (int c = 0; c < kMaxDim; ++c) {
          const float ph = __builtin_fmaf(half_eps, f[c], p[c]);
          p[c] = ph;
          const float xn = __builtin_fmaf(has_mass ? eps / m_safe[c] : eps, ph, x[c]);
          x[c] = c < dim ? xn : 0.0f;
        }
      }
#include "mlp_eval_body.inc"
      if (mode == 0) {  // H0 and the first (clamped) force
        h0 = clamp_nanprop(energy, -1e10f, 1e10f) + kinetic(p);
#pragma unroll
        for (int c = 0; c < kMaxDim; ++c) f[c] = clamp_nanprop(-g[c], -1e6f, 1e6f);
        e_last = energy;
        mode = 1;
        ++done;
      } else if (mode == 1) {
        // E finite => x finite and the gradient free of NaN (hmc_kernel.h); decided per WAVE because the
        // literal path re-runs the MFMA evaluation
        if (__all(__builtin_fabsf(energy) < __builtin_inff())) {
          float pz = 0.0f;
#pragma unroll
          for (int c = 0; c < kMaxDim; ++c) {
            const float fn = __builtin_amdgcn_fmed3f(-g[c], -1e6f, 1e6f);
            const float pn = __builtin_fmaf(half_eps, fn, p[c]);
            f[c] = fn;
            p[c] = pn;
            pz = __builtin_fmaf(pn, 0.0f, pz);
          }
          if (pz != pz) {  // momentum overflow: x is finite, so f stands
#pragma unroll
            for (int c = 0; c < kMaxDim; ++c) p[c] = nan_to_num0(p[c]);
          }
          e_last = energy;
          ++done;
        } else {  // literal semantics: NaN-propagating clamp, scrub, then re-evaluate on the scrubbed x
#pragma unroll
          for (int c = 0; c < kMaxDim; ++c) {
            const float fn = clamp_nanprop(-g[c], -1e6f, 1e6f);
            p[c] = nan_to_num0(__builtin_fmaf(half_eps, fn, p[c]));
            x[c] = nan_to_num0(x[c]);
          }
          mode = 2;
        }
      } else {  // mode 2: force and energy on the scrubbed position
#pragma unroll
        for (int c = 0; c < kMaxDim; ++c) f[c] = clamp_nanprop(-g[c], -1e6f, 1e6f);
        e_last = energy;
        mode = 1;
        ++done;
      }
    }
    const float h1 = clamp_nanprop(e_last, -1e10f, 1e10f) + kinetic(p);

    // ---- Metropolis accept (samplers/hmc.py:277-292)
    const float dlt = clamp_nanprop(h0 - h1, -50.0f, 50.0f);
    float acc_p = expf(dlt);
    acc_p = (acc_p > 1.0f) ? 1.0f : acc_p;  // clamp_(max=1); NaN stays NaN and rejects
    float uu;
    if (a.u) uu = active ? a.u[(int64_t)tr * a.n_chains + sample] : 2.0f;
    else uu = u01_half_open(pick(philox_at(a.key, (uint64_t)sample >> 2, a.step0 + 2ull * (uint64_t)tr + 1ull), (int)(sample & 3)));
    const bool accept = active && (uu < acc_p);
    if (accept) {
#pragma unroll
      for (int c = 0; c < kMaxDim; ++c) xc[c] = x[c];
    }
    const bool leader = active && h == 0;
    if (a.accept_mask && leader) a.accept_mask[(int64_t)tr * a.n_chains + sample] = accept ? 1 : 0;
    if (a.accept_count) {
      const unsigned long long b = __ballot(accept && leader);
      if (lane == 0 && b) atomicAdd(a.accept_count + tr, (uint32_t)__popcll(b));
    }
    if (a.traj && --until_keep == 0) {
      until_keep = a.thin;
      if (leader)
        for (int c = 0; c < dim; ++c) a.traj[sample * (int64_t)a.n_kept * dim + keep_off + c] = xc[c];
      keep_off += dim;
    }
  }
  if (active && h == 0)
    for (int c = 0; c < dim; ++c) a.x[sample * dim + c] = xc[c];
}

size_t mlp_smem_bytes() { return (size_t)(H * kW2Stride + H * 8 + 2 * H) * sizeof(float); }

// the two-moons shape this file specialises (layer 1 on the VALU); everything else: mlp_wide.hip
bool mlp_small(const ebm_energy_t& e, int32_t dim) { return e.n_comp == H && dim >= 1 && dim <= kMaxDim; }

int mlp_check(const ebm_energy_t& e, int32_t dim, const char* who, bool small_only) {
  if (!e.dev0) return fail(EBM_EINVAL, "%s: packed MLP parameters pointer is NULL", who);
  if (mlp_small(e, dim)) return 0;
  if (small_only)
    return fail(EBM_EDIM, "%s: HMC on the fused MLP energy supports hidden width 64 / 128 / 256 and 1 <= dim <= 128 (got %d, %d)",
                who, e.n_comp, dim);
  if (!mlp_wide_supported(e.n_comp, dim))
    return fail(EBM_EDIM, "%s: the fused MLP energy supports hidden width 64, 128 or 256 and 1 <= dim <= 128 (got %d, %d)", who, e.n_comp, dim);
  return 0;
}

int mlp_launch(const MlpArgs& a, hipStream_t st, const char* who) {
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  const size_t smem = mlp_smem_bytes();
  if (attr_once.first()) {  // > 64 KiB of dynamic LDS needs the opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_langevin_chain_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int64_t blocks = ceil_div64(a.n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "%s: too many chains for one launch", who);
  hipLaunchKernelGGL(mlp_langevin_chain_kernel, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch(who);
}

}  // namespace

int launch_langevin_chain_mlp(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                              float eta, float sqrt_eta, float noise_coef, const float* coef_table, int clamp_on,
                              float cmin, float cmax, int32_t thin, float* traj, const float* noise, uint64_t seed,
                              uint64_t offset, hipStream_t st) {
  const char* who = "ebm_langevin_chain_f32";
  if (int r = mlp_check(e, dim, who, false)) return r;
  if (!mlp_small(e, dim))
    return launch_mlp_wide(e.n_comp, e.dev0, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin, cmax,
                           thin, traj, noise, seed, offset, nullptr, nullptr, st, who);
  MlpArgs a;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.params = e.dev0; a.energy_out = nullptr; a.grad_out = nullptr;
  return mlp_launch(a, st, who);
}

int launch_hmc_chain_mlp(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog,
                         float eps, const float* eps_table, int32_t mass_kind, double mass_scalar, const float* mass_diag,
                         int32_t thin, float* traj, uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise,
                         const float* u, uint64_t seed, uint64_t offset, hipStream_t st) {
  const char* who = "ebm_hmc_chain_f32";
  if (!e.dev0) return fail(EBM_EINVAL, "%s: packed MLP parameters pointer is NULL", who);
  if (!mlp_small(e, dim) && mlp_wide_hmc_supported(e.n_comp, dim))  // the wide evaluation inside the same state machine
    return launch_hmc_chain_mlp_wide(e.n_comp, e.dev0, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar,
                                     mass_diag, thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, st, who);
  if (int r = mlp_check(e, dim, who, true)) return r;
  MlpHmcArgs a;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.n_mh = n_mh; a.n_leapfrog = n_leapfrog;
  a.eps = eps; a.eps_table = eps_table; a.mass_kind = mass_kind;
  a.mass_raw = (float)mass_scalar;
  a.mass_sqrt = (float)sqrt(mass_scalar);
  a.mass_safe = (float)(mass_scalar < 1e-10 ? 1e-10 : mass_scalar);
  a.mass_diag = mass_diag; a.thin = thin; a.n_kept = n_mh / thin; a.traj = traj;
  a.accept_mask = accept_mask; a.accept_count = accept_count; a.p_noise = p_noise; a.u = u;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.params = e.dev0;
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  const size_t smem = mlp_smem_bytes();
  if (attr_once.first()) {  // > 64 KiB of dynamic LDS needs the opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_hmc_chain_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int64_t blocks = ceil_div64(n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "%s: too many chains for one launch", who);
  hipLaunchKernelGGL(mlp_hmc_chain_kernel, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch(who);
}

int launch_energy_grad_mlp(const ebm_energy_t& e, const float* x, int64_t n_chains, int32_t dim, float* e_out,
                           float* g_out, hipStream_t st) {
  const char* who = "ebm_energy_grad_f32";
  if (int r = mlp_check(e, dim, who, false)) return r;
  if (!mlp_small(e, dim))
    return launch_mlp_wide(e.n_comp, e.dev0, const_cast<float*>(x), n_chains, dim, 0, 0.0f, 0.0f, 0.0f, nullptr, 0, 0.0f, 0.0f, 1,
                           nullptr, nullptr, 0, 0, e_out, g_out, st, who);
  MlpArgs a{};
  a.x = const_cast<float*>(x); a.n_chains = n_chains; a.dim = dim; a.k_steps = 0;
  a.thin = 1; a.n_kept = 0; a.params = e.dev0; a.energy_out = e_out; a.grad_out = g_out;
  return mlp_launch(a, st, who);
}

}  // namespace ebm
