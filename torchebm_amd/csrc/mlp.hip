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
    // ------------------------------------------------------------ forward: a2^T tiles
    f32x16 acc[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    // software-pipelined: the LDS operands of K-step s+1 are requested before the MFMAs of K-step s
    // issue, so their latency hides under 4 x 64 matrix-pipe cycles
    float w2a[kTiles], w2b[kTiles];
    float4 w1a = *reinterpret_cast<const float4*>(W1s + h * 8), w1b = w1a;
    float b1a = W1s[h * 8 + 4], b1b = b1a;
#pragma unroll
    for (int t = 0; t < kTiles; ++t) w2a[t] = W2s[(t * 32 + m) * kW2Stride + h];
    for (int s = 0; s < H / 2; ++s) {
      const int in = 2 * (s + 1 < H / 2 ? s + 1 : s) + h;  // next K index (clamped on the last step)
      w1b = *reinterpret_cast<const float4*>(W1s + in * 8);
      b1b = W1s[in * 8 + 4];
#pragma unroll
      for (int t = 0; t < kTiles; ++t) w2b[t] = W2s[(t * 32 + m) * kW2Stride + in];
      float a1 = b1a;
      a1 = __builtin_fmaf(w1a.x, x[0], a1);
      a1 = __builtin_fmaf(w1a.y, x[1], a1);
      a1 = __builtin_fmaf(w1a.z, x[2], a1);
      a1 = __builtin_fmaf(w1a.w, x[3], a1);
      const float h1 = a1 * sigmoidf_fast(a1);  // B[k = h][m]
#pragma unroll
      for (int t = 0; t < kTiles; ++t)           // A[row = m][k = h] = W2[j = 32t + m][i]
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(w2a[t], h1, acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // keep the issue order: next loads, this step's MFMAs
      w1a = w1b; b1a = b1b;
#pragma unroll
      for (int t = 0; t < kTiles; ++t) w2a[t] = w2b[t];
    }
    // ------------------------------------------------------------ energy, d2 = w3 * silu'(a2)
    float e_part = 0.0f;
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = t * 32 + row_of(r, h);
        const float a2 = acc[t][r] + b2s[j];
        const float sg = sigmoidf_fast(a2);
        const float w3 = w3s[j];
        e_part = __builtin_fmaf(w3, a2 * sg, e_part);
        acc[t][r] = w3 * (sg * (1.0f + a2 * (1.0f - sg)));
      }
    // ------------------------------------------------------------ backward: T^T tiles
    f32x16 tac[kTiles];
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) tac[t][r] = 0.0f;
    {
      float wa[kTiles], wb[kTiles];
#pragma unroll
      for (int t = 0; t < kTiles; ++t) wa[t] = W2s[row_of(0, h) * kW2Stride + t * 32 + m];
#pragma unroll
      for (int s = 0; s < 16 * kTiles; ++s) {     // K-step s: tile jt = s >> 4, register r = s & 15
        if (s + 1 < 16 * kTiles) {
          const int jn = ((s + 1) >> 4) * 32 + row_of((s + 1) & 15, h);
#pragma unroll
          for (int t = 0; t < kTiles; ++t) wb[t] = W2s[jn * kW2Stride + t * 32 + m];
        }
        const float d2 = acc[s >> 4][s & 15];     // B[k = h][m]: the K index j this half holds in register s & 15
#pragma unroll
        for (int t = 0; t < kTiles; ++t)          // A[row = m][k = h] = W2[j][i = 32t + m]
          tac[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[t], d2, tac[t], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < kTiles; ++t) wa[t] = wb[t];
      }
    }
    // ------------------------------------------------------------ g = W1^T (T^T * silu'(a1))
    float g[kMaxDim] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int t = 0; t < kTiles; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = t * 32 + row_of(r, h);
        const float4 w1 = *reinterpret_cast<const float4*>(W1s + i * 8);
        float a1 = W1s[i * 8 + 4];
        a1 = __builtin_fmaf(w1.x, x[0], a1);
        a1 = __builtin_fmaf(w1.y, x[1], a1);
        a1 = __builtin_fmaf(w1.z, x[2], a1);
        a1 = __builtin_fmaf(w1.w, x[3], a1);
        const float sg = sigmoidf_fast(a1);
        const float d1 = tac[t][r] * (sg * (1.0f + a1 * (1.0f - sg)));
        g[0] = __builtin_fmaf(w1.x, d1, g[0]);
        g[1] = __builtin_fmaf(w1.y, d1, g[1]);
        g[2] = __builtin_fmaf(w1.z, d1, g[2]);
        g[3] = __builtin_fmaf(w1.w, d1, g[3]);
      }
#pragma unroll
    for (int c = 0; c < kMaxDim; ++c) g[c] += __shfl_xor(g[c], 32);  // the two K-halves of a sample
    const float energy = e_part + __shfl_xor(e_part, 32) + b3;

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

size_t mlp_smem_bytes() { return (size_t)(H * kW2Stride + H * 8 + 2 * H) * sizeof(float); }

int mlp_check(const ebm_energy_t& e, int32_t dim, const char* who) {
  if (e.n_comp != H) return fail(EBM_EDIM, "%s: the fused MLP energy supports hidden width %d (got %d)", who, H, e.n_comp);
  if (dim < 1 || dim > kMaxDim) return fail(EBM_EDIM, "%s: the fused MLP energy supports 1 <= dim <= %d (got %d)", who, kMaxDim, dim);
  if (!e.dev0) return fail(EBM_EINVAL, "%s: packed MLP parameters pointer is NULL", who);
  return 0;
}

int mlp_launch(const MlpArgs& a, hipStream_t st, const char* who) {
  static bool attr_set = false;
  const size_t smem = mlp_smem_bytes();
  if (!attr_set) {  // > 64 KiB of dynamic LDS needs the opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_langevin_chain_kernel),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
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
  if (int r = mlp_check(e, dim, who)) return r;
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

int launch_energy_grad_mlp(const ebm_energy_t& e, const float* x, int64_t n_chains, int32_t dim, float* e_out,
                           float* g_out, hipStream_t st) {
  const char* who = "ebm_energy_grad_f32";
  if (int r = mlp_check(e, dim, who)) return r;
  MlpArgs a{};
  a.x = const_cast<float*>(x); a.n_chains = n_chains; a.dim = dim; a.k_steps = 0;
  a.thin = 1; a.n_kept = 0; a.params = e.dev0; a.energy_out = e_out; a.grad_out = g_out;
  return mlp_launch(a, st, who);
}

}  // namespace ebm
