// The PARAMETER gradients of a training step through an EBM_ENERGY_MLP network, from the activation planes
// ebm_mlp_backward_acts_f32 stored (ebm_mlp_param_grads_f32, ABI 7).  What autograd does for loss.backward() through the two
// nn.Linear weights of examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-31 (torchebm/losses/contrastive_divergence.py:128-155).
//
//   dW2 = sum_k s_k d2[:,k] h1[:,k]^T   db2 = sum_k s_k d2[:,k]   dW1 = sum_k s_k d1[:,k] x[k,:]   db1 = sum_k s_k d1[:,k]
//   dw3 = sum_k s_k h2[:,k]             db3 = sum_k s_k
//
// are small-output products over K = n rows (131 072 for BASELINE config 5): 4 GFLOP next to 268 MB of planes -- an HBM-bound pass
// if the planes are read ONCE.  The library route reads them 2.2 times in six launches (two row-block batched GEMMs at 2.5 TB/s,
// four row reductions); this kernel reads every plane once:
//   * one workgroup per CU walks chunks of KC = 32 rows (columns of the hidden-major planes): global -> registers (seed scaling,
//     the row sums) -> LDS, two stages, one barrier per chunk;
//   * the products run on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation): wave w owns a quadrant of dW2 and row
//     tile w of dW1; both operands of a K-step come from the same [row][k] walk over an LDS tile (one ds_read_b128 = four K-steps);
//     a K-step's two K indices are 8 t + e and 8 t + 4 + e -- any pairing works as long as both operands use it;
//   * every workgroup writes ONE partial record (the packed parameter order, W1 padded to 32 DT columns), and a second small kernel
//     adds the records in a fixed order: the result does not depend on scheduling (the training step stays bit-reproducible).
#include "ebm_common.h"

namespace ebm {
namespace mlpgrads {

typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kThreads = 256, KC = 32, PITCH = KC + 4;

template <int HT, int DT>
struct Shape {
  static constexpr int H = 32 * HT, DP = 32 * DT, XP = DP + 1;
  static constexpr int plane_floats = H * PITCH;
  static constexpr int stage_floats = 3 * plane_floats + KC * XP;
  static constexpr size_t smem_bytes = (size_t)2 * stage_floats * sizeof(float);
  // one partial record: W1 [H][DP] | b1 [H] | W2 [H][H] | b2 [H] | w3 [H] | b3
  static constexpr int off_w1 = 0, off_b1 = H * DP, off_w2 = off_b1 + H, off_b2 = off_w2 + H * H, off_w3 = off_b2 + H, off_b3 = off_w3 + H;
  static constexpr int record = off_b3 + 1;
};

struct Args {
  const float* acts;   // [4][H][stride]: h1 | h2 | d2 | d1
  int64_t stride;      // n rounded up to a multiple of 128
  const float* x;      // [n][dim]
  int64_t n;
  int32_t dim;
  const float* seed;   // [n] or NULL (= 1: the planes are already scaled)
  float* partials;     // [gridDim.x][record]
  int64_t chunks;      // stride / KC
};

extern __shared__ __attribute__((aligned(16))) float grads_smem[];

template <int HT, int DT>
__global__ __launch_bounds__(kThreads) void mlp_param_grads_kernel(Args a) {
  using S = Shape<HT, DT>;
  constexpr int H = S::H, DP = S::DP, XP = S::XP;
  constexpr int XI = (KC * DP + kThreads - 1) / kThreads;  // x elements a thread stages per chunk
  constexpr int RT = HT / 2, CT = HT / 2;                  // dW2 tiles of a wave: RT x CT (HT = 4: a 64 x 64 quadrant; HT = 2: one tile)
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = lane & 31, hf = lane >> 5;
  const int c4 = t & 7, r0 = t >> 3;
  const int dim = a.dim;
  const int64_t G = gridDim.x;
  const auto stage = [](int i) { return grads_smem + i * S::stage_floats; };

  // x staging: element idx = t + 256 i of the chunk's [KC][dim] block (fixed per thread)
  int xrow[XI], xcol[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = t + kThreads * i;
    xrow[i] = idx < KC * dim ? idx / dim : -1;
    xcol[i] = idx < KC * dim ? idx - (idx / dim) * dim : 0;
  }
  for (int i = t; i < KC * XP; i += kThreads) {  // columns dim .. DP - 1 stay zero
    stage(0)[3 * S::plane_floats + i] = 0.0f;
    stage(1)[3 * S::plane_floats + i] = 0.0f;
  }

  float4 rh1[HT], rh2[HT], rd2[HT], rd1[HT], rs;
  float xr[XI];
  float sum_d2[HT], sum_d1[HT], sum_h2[HT], sum_seed = 0.0f;
#pragma unroll
  for (int j = 0; j < HT; ++j) sum_d2[j] = sum_d1[j] = sum_h2[j] = 0.0f;

  const auto load_chunk = [&](int64_t c) __attribute__((always_inline)) {
    const int64_t col = c * KC + 4 * c4;
    const float* p = a.acts + (int64_t)r0 * a.stride + col;
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      const int64_t row = (int64_t)32 * j * a.stride;
      rh1[j] = *reinterpret_cast<const float4*>(p + row);
      rh2[j] = *reinterpret_cast<const float4*>(p + row + (int64_t)H * a.stride);
      rd2[j] = *reinterpret_cast<const float4*>(p + row + (int64_t)2 * H * a.stride);
      rd1[j] = *reinterpret_cast<const float4*>(p + row + (int64_t)3 * H * a.stride);
    }
    // (rows n .. stride - 1: the planes hold zeros there, and the seed is zero too -- db3 counts real rows only)
    rs.x = col + 0 < a.n ? (a.seed ? a.seed[col + 0] : 1.0f) : 0.0f;
    rs.y = col + 1 < a.n ? (a.seed ? a.seed[col + 1] : 1.0f) : 0.0f;
    rs.z = col + 2 < a.n ? (a.seed ? a.seed[col + 2] : 1.0f) : 0.0f;
    rs.w = col + 3 < a.n ? (a.seed ? a.seed[col + 3] : 1.0f) : 0.0f;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int64_t k = c * KC + xrow[i];
      xr[i] = (xrow[i] >= 0 && k < a.n) ? a.x[k * dim + xcol[i]] : 0.0f;
    }
  };
  const auto mul4 = [](float4 v, float4 s) { return make_float4(v.x * s.x, v.y * s.y, v.z * s.z, v.w * s.w); };
  const auto add4 = [](float4 v) { return (v.x + v.y) + (v.z + v.w); };
  const auto store_chunk = [&](float* st) __attribute__((always_inline)) {
    if (r0 == 0) sum_seed += add4(rs);
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      const float4 d2 = mul4(rd2[j], rs), d1 = mul4(rd1[j], rs), h2 = mul4(rh2[j], rs);
      sum_d2[j] += add4(d2);
      sum_d1[j] += add4(d1);
      sum_h2[j] += add4(h2);
      const int o = (r0 + 32 * j) * PITCH + 4 * c4;
      *reinterpret_cast<float4*>(st + o) = d2;
      *reinterpret_cast<float4*>(st + S::plane_floats + o) = rh1[j];
      *reinterpret_cast<float4*>(st + 2 * S::plane_floats + o) = d1;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i)
      if (xrow[i] >= 0) st[3 * S::plane_floats + xrow[i] * XP + xcol[i]] = xr[i];
  };

  f32x16 acc2[RT][CT], acc1[DT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc2[i][j] = (f32x16)(0.0f);
#pragma unroll
  for (int j = 0; j < DT; ++j) acc1[j] = (f32x16)(0.0f);
  const int rt0 = (w >> 1) * RT, ct0 = (w & 1) * CT;
  const bool has_w1 = w < HT;  // row tile w of dW1

  const auto compute = [&](const float* st) __attribute__((always_inline)) {
    const float* d2s = st;
    const float* h1s = st + S::plane_floats;
    const float* d1s = st + 2 * S::plane_floats;
    const float* xs = st + 3 * S::plane_floats;
#pragma unroll
    for (int tt = 0; tt < KC / 8; ++tt) {
      const int ko = 8 * tt + 4 * hf;
      float4 av[RT], bv[CT];
#pragma unroll
      for (int i = 0; i < RT; ++i) av[i] = *reinterpret_cast<const float4*>(d2s + (32 * (rt0 + i) + r) * PITCH + ko);
#pragma unroll
      for (int j = 0; j < CT; ++j) bv[j] = *reinterpret_cast<const float4*>(h1s + (32 * (ct0 + j) + r) * PITCH + ko);
      float4 a1 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      float xb[4][DT];
      if (has_w1) {
        a1 = *reinterpret_cast<const float4*>(d1s + (32 * w + r) * PITCH + ko);
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int j = 0; j < DT; ++j) xb[e][j] = xs[(ko + e) * XP + 32 * j + r];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int j = 0; j < CT; ++j)
            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(((const float*)&av[i])[e], ((const float*)&bv[j])[e], acc2[i][j], 0, 0, 0);
        if (has_w1) {
#pragma unroll
          for (int j = 0; j < DT; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(((const float*)&a1)[e], xb[e][j], acc1[j], 0, 0, 0);
        }
      }
    }
  };

  int64_t c = blockIdx.x;
  load_chunk(c);
  __syncthreads();  // the zeroed x columns
  store_chunk(stage(0));
  __syncthreads();
  for (int it = 0;; ++it) {
    const int64_t cn = c + G;
    const bool more = cn < a.chunks;
    if (more) load_chunk(cn);
    compute(stage(it & 1));
    if (!more) break;
    store_chunk(stage((it + 1) & 1));
    __syncthreads();
    c = cn;
  }

  // ---- this workgroup's partial record
  float* P = a.partials + (int64_t)blockIdx.x * S::record;
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 32 * (rt0 + i) + 8 * (q >> 2) + 4 * hf + (q & 3), col = 32 * (ct0 + j) + r;
        P[S::off_w2 + row * H + col] = acc2[i][j][q];
      }
  if (has_w1) {
#pragma unroll
    for (int j = 0; j < DT; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = 32 * w + 8 * (q >> 2) + 4 * hf + (q & 3), col = 32 * j + r;
        P[S::off_w1 + row * DP + col] = acc1[j][q];
      }
  }
  // the row sums: eight consecutive lanes hold the pieces of a row
#pragma unroll
  for (int j = 0; j < HT; ++j) {
#pragma unroll
    for (int m = 1; m < 8; m <<= 1) {
      sum_d2[j] += __shfl_xor(sum_d2[j], m);
      sum_d1[j] += __shfl_xor(sum_d1[j], m);
      sum_h2[j] += __shfl_xor(sum_h2[j], m);
    }
    if (c4 == 0) {
      P[S::off_b2 + r0 + 32 * j] = sum_d2[j];
      P[S::off_b1 + r0 + 32 * j] = sum_d1[j];
      P[S::off_w3 + r0 + 32 * j] = sum_h2[j];
    }
  }
#pragma unroll
  for (int m = 1; m < 8; m <<= 1) sum_seed += __shfl_xor(sum_seed, m);
  if (t == 0) P[S::off_b3] = sum_seed;
}

// out[packed parameter order, W1 unpadded] = sum over the G partial records, in a fixed order: 32 elements x 8 groups of records per
// workgroup (group q adds records q, q + 8, ...), then the eight group sums in order.
template <int HT, int DT>
__global__ __launch_bounds__(256) void mlp_param_grads_reduce_kernel(const float* __restrict__ partials, int32_t G, int32_t dim, float* __restrict__ out) {
  using S = Shape<HT, DT>;
  __shared__ float part[8][32];
  const int e = blockIdx.x * 32 + (threadIdx.x & 31), q = threadIdx.x >> 5;
  float s = 0.0f;
  if (e < S::record)
    for (int g = q; g < G; g += 8) s += partials[(int64_t)g * S::record + e];
  part[q][threadIdx.x & 31] = s;
  __syncthreads();
  if (q == 0 && e < S::record) {
    float v = part[0][threadIdx.x];
#pragma unroll
    for (int i = 1; i < 8; ++i) v += part[i][threadIdx.x];
    if (e < S::off_b1) {
      const int row = e / S::DP, col = e - row * S::DP;
      if (col < dim) out[row * dim + col] = v;
    } else {
      out[e - S::off_b1 + S::H * dim] = v;
    }
  }
}

int grid_of(int64_t chunks) {
  int dev = 0, cus = 0;
  (void)hipGetDevice(&dev);
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  return (int)(chunks < cus ? chunks : cus);
}

template <int HT, int DT>
int launch(const Args& a0, float* out, hipStream_t st, const char* who) {
  using S = Shape<HT, DT>;
  static DeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_param_grads_kernel<HT, DT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)S::smem_bytes);
  Args a = a0;
  const int G = grid_of(a.chunks);
  hipLaunchKernelGGL((mlp_param_grads_kernel<HT, DT>), dim3(G), dim3(kThreads), S::smem_bytes, st, a);
  if (int r = check_launch(who)) return r;
  hipLaunchKernelGGL((mlp_param_grads_reduce_kernel<HT, DT>), dim3((S::record + 31) / 32), dim3(256), 0, st, a.partials, G, a.dim, out);
  return check_launch(who);
}

}  // namespace mlpgrads

// floats of workspace ebm_mlp_param_grads_f32 needs on the current device (one partial record per workgroup, one workgroup per CU)
int64_t mlp_param_grads_work_floats(int32_t hidden, int32_t dim, int64_t n) {
  const int64_t chunks = (n + 127) / 128 * 128 / mlpgrads::KC;
  const int64_t dp = 32 * ((dim + 31) / 32);
  const int64_t record = (int64_t)hidden * dp + hidden + (int64_t)hidden * hidden + 2 * hidden + 1;
  return record * mlpgrads::grid_of(chunks > 0 ? chunks : 1);
}

int launch_mlp_param_grads(int32_t hidden, const float* acts, const float* x, int64_t n, int32_t dim, const float* seed, float* work,
                           float* out, hipStream_t st, const char* who) {
  mlpgrads::Args a{};
  a.acts = acts; a.stride = (n + 127) / 128 * 128; a.x = x; a.n = n; a.dim = dim; a.seed = seed; a.partials = work;
  a.chunks = a.stride / mlpgrads::KC;
  const int dt = (dim + 31) / 32;
  if (hidden == 64) return dt == 1 ? mlpgrads::launch<2, 1>(a, out, st, who) : mlpgrads::launch<2, 2>(a, out, st, who);
  return dt == 1 ? mlpgrads::launch<4, 1>(a, out, st, who) : mlpgrads::launch<4, 2>(a, out, st, who);
}

}  // namespace ebm
