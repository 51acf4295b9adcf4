// k-fused Langevin chain (and stand-alone energy / gradient) for the two-hidden-layer SiLU MLP energy
//     E(x) = w3 . silu(W2 silu(W1 x + b1) + b2) + b3,      x in R^D,  D <= 128,  hidden width H in {64, 128}
// -- the reference's benchmark MLP (benchmarks/registry.py:372-387: Linear(dim, 128) ... at dim 8 / 32 / 128) beyond the
// 2-D two-moons network that mlp.hip specialises (there the first layer is two FMAs per hidden unit on the VALU).
//
// All FOUR contractions of one evaluation run on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32), and every
// operand that is not a weight is used where the previous contraction left it:
//   a1^T[i, m] = sum_c W1[i, c] x[c, m]        h1 = silu(a1 + b1)       s1 = silu'(a1 + b1)
//   a2^T[j, m] = sum_i W2[j, i] h1[i, m]       E += w3[j] silu(a2 + b2) d2 = w3[j] silu'(a2 + b2)
//   T^T[i, m]  = sum_j W2[j, i] d2[j, m]       d1 = T * s1
//   g^T[c, m]  = sum_i W1[i, c] d1[i, m]
// A wavefront owns 32 chains: lane l = (m, h), m = l & 31 the chain, h = l >> 5 the K-half of the MFMA.  Everything is
// kept TRANSPOSED (rows = feature index, columns = chain) in the 32x32 C/D layout -- register r of lane (m, h) holds
// row (r & 3) + 8 (r >> 2) + 4 h of a 32-row tile -- INCLUDING THE STATE x: the K index of every contraction is
// enumerated in the order that layout holds it (K-step (tile, r): half h contributes k = 32 tile + row_of(r, h)), so
// the B operand of each MFMA is simply register r of the previous result, the gradient comes out in the layout the
// state lives in, and no value ever changes lanes.  The A operands are single LDS words: W1 with row stride
// Dpad + 1 and W2 with row stride H + 1 are conflict-free for both the row walk (forward) and the column walk
// (backward).  Registers: x, s1, d2, T and g are 16 H/32 (or 16 Dpad/32) values each -- up to 320 at D = H = 128 --
// so the kernel runs one wave per SIMD on the unified 512-entry register file (accumulators in AGPRs); the LDS
// (W1 + W2 = 132 KiB at D = H = 128) allows one workgroup per CU anyway.
// Reference: torchebm/samplers/langevin_dynamics.py:154-185 (the loop), core/base_integrator.py:711-731 (the update).
#include "ebm_common.h"

namespace ebm {
namespace {

constexpr int kBlock = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct WideArgs {
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

extern __shared__ __attribute__((aligned(16))) float wide_smem[];

__device__ __forceinline__ float sigmoid_fast(float a) { return __builtin_amdgcn_rcpf(1.0f + __expf(-a)); }
__device__ __forceinline__ constexpr int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// out[t] += A_t * B over NK x 16 K-steps; the LDS operands of step s + 1 are requested before the MFMAs of step s
// issue, so their latency hides under NT x 64 matrix-pipe cycles.  addr(t, tk, r): LDS word of A for output tile t at
// K-step (tk, r) (this lane's row / K-half folded in by the caller); bval(tk, r): this lane's B value.
template <int NT, int NK, class Addr, class Bval>
__device__ __forceinline__ void contract(f32x16 (&out)[NT], const float* lds, Addr addr, Bval bval) {
  float cur[NT], nxt[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) cur[t] = lds[addr(t, 0, 0)];
#pragma unroll
  for (int tk = 0; tk < NK; ++tk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s = tk * 16 + r;
      if (s + 1 < NK * 16) {
#pragma unroll
        for (int t = 0; t < NT; ++t) nxt[t] = lds[addr(t, (s + 1) >> 4, (s + 1) & 15)];
      }
      const float b = bval(tk, r);
#pragma unroll
      for (int t = 0; t < NT; ++t) out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t], b, out[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // keep the issue order: next loads, this step's MFMAs
#pragma unroll
      for (int t = 0; t < NT; ++t) cur[t] = nxt[t];
    }
}

template <int HT, int DT>
__global__ __launch_bounds__(kBlock, 1) void mlp_wide_chain_kernel(WideArgs a) {
  constexpr int H = 32 * HT, DP = 32 * DT;  // hidden width, padded input width
  constexpr int S1 = DP + 1, S2 = H + 1;    // LDS row strides
  float* W2s = wide_smem;                   // [H][S2]
  float* W1s = W2s + H * S2;                // [H][S1], columns >= dim zero
  float* b1s = W1s + H * S1;                // [H]
  float* b2s = b1s + H;
  float* w3s = b2s + H;
  const int dim = a.dim;
  {  // stage the weights (once per launch)
    const float* W1g = a.params;
    const float* b1g = W1g + H * dim;
    const float* W2g = b1g + H;
    const float* b2g = W2g + H * H;
    const float* w3g = b2g + H;
    for (int i = threadIdx.x; i < H * H; i += kBlock) W2s[(i / H) * S2 + (i % H)] = W2g[i];
    for (int i = threadIdx.x; i < H * DP; i += kBlock) {
      const int row = i / DP, c = i - row * DP;
      W1s[row * S1 + c] = c < dim ? W1g[row * dim + c] : 0.0f;
    }
    for (int i = threadIdx.x; i < H; i += kBlock) {
      b1s[i] = b1g[i];
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
  const bool quads = (dim & 3) == 0;  // a register quad r = 4q .. 4q+3 is four consecutive, 16-byte aligned columns

  // the state in the C/D layout: xr[td][r] = x[sample][32 td + row_of(r, h)], zero beyond dim
  float xr[DT][16];
#pragma unroll
  for (int td = 0; td < DT; ++td)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = 32 * td + 8 * q + 4 * h;
      if (quads && active && c0 + 3 < dim) {
        const float4 v = *reinterpret_cast<const float4*>(a.x + sample * dim + c0);
        xr[td][4 * q] = v.x; xr[td][4 * q + 1] = v.y; xr[td][4 * q + 2] = v.z; xr[td][4 * q + 3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) xr[td][4 * q + i] = (active && c0 + i < dim) ? a.x[sample * dim + c0 + i] : 0.0f;
      }
    }

  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  const int n_evals = a.k_steps > 0 ? a.k_steps : 1;

  for (int step = 0; step < n_evals; ++step) {
    // ------------------------------------------------------------ layer 1: a1^T tiles, K = input columns
    f32x16 u[HT];  // a1, then h1 = silu(a1)
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) u[t][r] = 0.0f;
    contract<HT, DT>(u, W1s, [&](int t, int tk, int r) { return (32 * t + m) * S1 + 32 * tk + row_of(r, h); },
                     [&](int tk, int r) { return xr[tk][r]; });
    f32x16 s1[HT];  // silu'(a1)
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float a1 = u[t][r] + b1s[32 * t + row_of(r, h)];
        const float sg = sigmoid_fast(a1);
        u[t][r] = a1 * sg;
        s1[t][r] = sg * (1.0f + a1 * (1.0f - sg));
      }
    // ------------------------------------------------------------ layer 2: a2^T tiles, K = hidden units of layer 1
    f32x16 v[HT];  // a2, then d2 = w3 * silu'(a2)
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) v[t][r] = 0.0f;
    contract<HT, HT>(v, W2s, [&](int t, int tk, int r) { return (32 * t + m) * S2 + 32 * tk + row_of(r, h); },
                     [&](int tk, int r) { return u[tk][r]; });
    float e_part = 0.0f;
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * t + row_of(r, h);
        const float a2 = v[t][r] + b2s[j];
        const float sg = sigmoid_fast(a2);
        const float w3 = w3s[j];
        e_part = __builtin_fmaf(w3, a2 * sg, e_part);
        v[t][r] = w3 * (sg * (1.0f + a2 * (1.0f - sg)));
      }
    // ------------------------------------------------------------ backward through W2: T^T tiles, K = units of layer 2
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) u[t][r] = 0.0f;
    contract<HT, HT>(u, W2s, [&](int t, int tk, int r) { return (32 * tk + row_of(r, h)) * S2 + 32 * t + m; },
                     [&](int tk, int r) { return v[tk][r]; });
#pragma unroll
    for (int t = 0; t < HT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) u[t][r] *= s1[t][r];  // d1
    // ------------------------------------------------------------ backward through W1: g^T tiles, K = units of layer 1
    f32x16 g[DT];
#pragma unroll
    for (int t = 0; t < DT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) g[t][r] = 0.0f;
    contract<DT, HT>(g, W1s, [&](int t, int tk, int r) { return (32 * tk + row_of(r, h)) * S1 + 32 * t + m; },
                     [&](int tk, int r) { return u[tk][r]; });
    const float energy = e_part + __shfl_xor(e_part, 32) + b3;  // the two K-halves hold the two halves of the rows

    if (a.k_steps == 0) {  // evaluation only
      if (active) {
        if (a.energy_out && h == 0) a.energy_out[sample] = energy;
        if (a.grad_out) {
#pragma unroll
          for (int td = 0; td < DT; ++td)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int c = 32 * td + row_of(r, h);
              if (c < dim) a.grad_out[sample * dim + c] = g[td][r];
            }
        }
      }
      return;
    }

    // ------------------------------------------------------------ Euler-Maruyama update (reference op order)
    if (a.table) {
      const float4 tb = a.table[step];
      eta = tb.x; sqrt_eta = tb.y; noise_coef = tb.z;
    }
#pragma unroll
    for (int td = 0; td < DT; ++td)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * td + 8 * q + 4 * h;
        float eps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (c0 < dim) {
          if (a.noise) {
            if (active)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (c0 + i < dim) eps[i] = a.noise[((int64_t)step * a.n_chains + sample) * dim + c0 + i];
          } else if (quads) {  // the quad is exactly one Philox counter
            const F4 nrm = normal4_at(a.key, ((uint64_t)sample * (uint64_t)dim + (uint64_t)c0) >> 2, a.step0 + (uint64_t)step);
#pragma unroll
            for (int i = 0; i < 4; ++i) eps[i] = nrm.v[i];
          } else {
            uint64_t have = ~0ull;
            F4 nrm;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint64_t e = (uint64_t)sample * (uint64_t)dim + (uint64_t)(c0 + i);
              if ((e >> 2) != have) {
                have = e >> 2;
                nrm = normal4_at(a.key, have, a.step0 + (uint64_t)step);
              }
              const int w = (int)(e & 3);
              eps[i] = w == 0 ? nrm.v[0] : (w == 1 ? nrm.v[1] : (w == 2 ? nrm.v[2] : nrm.v[3]));
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = 4 * q + i;
          const float x1 = xr[td][r] - eta * g[td][r];
          const float dw = eps[i] * sqrt_eta;
          float nv = x1 + noise_coef * dw;
          if (a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
          xr[td][r] = (c0 + i < dim) ? nv : 0.0f;
        }
      }
    if (a.traj && --until_keep == 0) {
      until_keep = a.thin;
      if (active) {
        float* dst = a.traj + sample * (int64_t)a.n_kept * dim + keep_off;
#pragma unroll
        for (int td = 0; td < DT; ++td)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = 32 * td + row_of(r, h);
            if (c < dim) dst[c] = xr[td][r];
          }
      }
      keep_off += dim;
    }
  }
  if (active) {
#pragma unroll
    for (int td = 0; td < DT; ++td)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * td + row_of(r, h);
        if (c < dim) a.x[sample * dim + c] = xr[td][r];
      }
  }
}

template <int HT, int DT>
int launch_one(const WideArgs& a, hipStream_t st, const char* who) {
  constexpr int H = 32 * HT, DP = 32 * DT;
  const size_t smem = (size_t)(H * (H + 1) + H * (DP + 1) + 3 * H) * sizeof(float);
  static bool attr_set = false;
  if (!attr_set) {  // > 64 KiB of dynamic LDS needs the opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_wide_chain_kernel<HT, DT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    attr_set = true;
  }
  const int64_t blocks = ceil_div64(a.n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "%s: too many chains for one launch", who);
  hipLaunchKernelGGL((mlp_wide_chain_kernel<HT, DT>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch(who);
}

}  // namespace

bool mlp_wide_supported(int32_t hidden, int32_t dim) { return (hidden == 64 || hidden == 128) && dim >= 1 && dim <= 128; }

// k_steps == 0: evaluation into energy_out / grad_out; else the k-fused chain
int launch_mlp_wide(int32_t hidden, const float* params, float* x, int64_t n_chains, int32_t dim, int32_t k_steps, float eta,
                    float sqrt_eta, float noise_coef, const float* coef_table, int clamp_on, float cmin, float cmax,
                    int32_t thin, float* traj, const float* noise, uint64_t seed, uint64_t offset, float* energy_out,
                    float* grad_out, hipStream_t st, const char* who) {
  WideArgs a;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = thin > 0 ? k_steps / thin : 0; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.params = params; a.energy_out = energy_out; a.grad_out = grad_out;
  const int dt = (dim + 31) / 32;
#define EBM_WIDE(HTV)                                    \
  switch (dt) {                                          \
    case 1: return launch_one<HTV, 1>(a, st, who);       \
    case 2: return launch_one<HTV, 2>(a, st, who);       \
    case 3: return launch_one<HTV, 3>(a, st, who);       \
    default: return launch_one<HTV, 4>(a, st, who);      \
  }
  if (hidden == 64) { EBM_WIDE(2) }
  EBM_WIDE(4)
#undef EBM_WIDE
}

}  // namespace ebm
