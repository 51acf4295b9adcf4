// The PARAMETER gradients of a training step through an EBM_ENERGY_MLP network, from the activation planes
// ebm_mlp_backward_acts_f32 stored (ebm_mlp_param_grads_f32, ABI 7).  What autograd does for loss.backward() through the two
// nn.Linear weights of examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-31 (torchebm/losses/contrastive_divergence.py:128-155).
//
//   dW2 = sum_k s_k d2[:,k] h1[:,k]^T   db2 = sum_k s_k d2[:,k]   dW1 = sum_k s_k d1[:,k] x[k,:]   db1 = sum_k s_k d1[:,k]
//   dw3 = sum_k s_k h2[:,k]             db3 = sum_k s_k          with h2 = silu(a2), d2 = w3 silu'(a2) recomputed from the a2 plane
//
// are small-output products over K = n rows (131 072 for BASELINE config 5): 4 GFLOP next to 201 MB of planes.  The library
// route read (four) planes 2.2 times in six launches (two row-block batched GEMMs at 2.5 TB/s, four row reductions); this kernel reads
// every plane once -- and is bound by neither the bytes nor the MFMAs but by the serial phases of a chunk (see the main loop):
//   * two workgroups per CU walk chunks of KC = 32 rows -- one tile of the planes, a contiguous 12 H-float block: every load
//     instruction of a wave is 1 KB of consecutive addresses --: global -> registers (h2 / d2 from a2, seed scaling, the row sums)
//     -> one LDS stage; while one workgroup multiplies the other loads and stages;
//   * the products run on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation): wave w owns a quadrant of dW2 and row
//     tile w of dW1; both operands of a K-step come from the same [row][k] walk over an LDS tile (one ds_read_b128 = four K-steps);
//     a K-step's two K indices are 8 t + e and 8 t + 4 + e -- any pairing works as long as both operands use it;
//   * every workgroup writes ONE partial record (the packed parameter order, W1 padded to 32 DT columns), and a second small kernel
//     adds the records in a fixed order: the result does not depend on scheduling (the training step stays bit-reproducible).
#include "ebm_common.h"

namespace ebm {
namespace mlpgrads {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));  // (a native vector: HIP's float4 struct is copied through private memory)
constexpr int kThreads = 256, KC = 32, PITCH = KC + 4;

template <int HT, int DT>
struct Shape {
  static constexpr int H = 32 * HT, DP = 32 * DT, XP = DP + 1;
  static constexpr int plane_floats = H * PITCH;
  static constexpr int stage_floats = 3 * plane_floats + KC * XP + 4;  // (+ 4: where the x staging slots a thread does not have land)
  static constexpr size_t smem_bytes = (size_t)stage_floats * sizeof(float);  // one stage: two workgroups share a CU
  // one partial record: W1 [H][DP] | b1 [H] | W2 [H][H] | b2 [H] | w3 [H] | b3
  static constexpr int off_w1 = 0, off_b1 = H * DP, off_w2 = off_b1 + H, off_b2 = off_w2 + H * H, off_w3 = off_b2 + H, off_b3 = off_w3 + H;
  static constexpr int record = off_b3 + 1;
};

struct Args {
  const float* acts;   // [stride / 32][3][H][32]: h1 | a2 | d1 of 32 rows each, hidden-major (ebm_mlp_backward_acts_f32)
  const float* w3;     // [H]: the last layer's weights (d2 = w3 silu'(a2))
  int64_t stride;      // n rounded up to a multiple of 128
  const float* x;      // [n][dim]
  int64_t n;
  int32_t dim;
  const float* seed;   // [n] or NULL (= 1)
  float* partials;     // [gridDim.x][record]
  int64_t chunks;      // stride / KC
};

extern __shared__ __attribute__((aligned(16))) float grads_smem[];

template <int HT, int DT>
__global__ __launch_bounds__(kThreads, 2) void mlp_param_grads_kernel(Args a) {
  using S = Shape<HT, DT>;
  constexpr int H = S::H, DP = S::DP, XP = S::XP;
  constexpr int XI = (KC * DP + kThreads - 1) / kThreads;  // x elements a thread stages per chunk
  constexpr int RT = HT / 2, CT = HT / 2;                  // dW2 tiles of a wave: RT x CT (HT = 4: a 64 x 64 quadrant; HT = 2: one tile)
  const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = lane & 31, hf = lane >> 5;
  const int c4 = t & 7, r0 = t >> 3;
  const int dim = a.dim;
  const int64_t G = gridDim.x;
  float* const stage = grads_smem;

  // x staging: element idx = t + 256 i of the chunk's [KC][dim] block (fixed per thread)
  int xrow[XI], xcol[XI], xoff[XI];
#pragma unroll
  for (int i = 0; i < XI; ++i) {
    const int idx = t + kThreads * i;
    xrow[i] = idx < KC * dim ? idx / dim : -1;
    xcol[i] = idx < KC * dim ? idx - (idx / dim) * dim : 0;
    xoff[i] = 3 * S::plane_floats + (xrow[i] >= 0 ? xrow[i] * XP + xcol[i] : KC * XP);  // (no slot: the spare word behind the block)
  }
  for (int i = t; i < KC * XP; i += kThreads) stage[3 * S::plane_floats + i] = 0.0f;  // columns dim .. DP - 1 stay zero

  struct Regs {  // one chunk on its way from global memory to LDS
    f32x4 h1[HT], a2[HT], d1[HT], s;
    float x[XI];
  };
  Regs rq;  // the chunk behind the one being multiplied
  float sum_d2[HT], sum_d1[HT], sum_h2[HT], sum_seed = 0.0f, w3r[HT];
#pragma unroll
  for (int j = 0; j < HT; ++j) {
    sum_d2[j] = sum_d1[j] = sum_h2[j] = 0.0f;
    w3r[j] = a.w3[r0 + 32 * j];  // this thread's rows of the planes
  }

  const auto load_chunk = [&](Regs& q, int64_t c) __attribute__((always_inline)) {
    const int64_t col = c * KC + 4 * c4;
    // chunk c = tile c of the planes: one contiguous [3][H][32] block; thread t reads floats 4 t .. 4 t + 3 of every 32-row slab
    // (nontemporal, like the stores that wrote them: read once)
    const float* p = a.acts + c * (int64_t)(3 * H * KC) + 4 * t;
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      q.h1[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 32 * j * KC));
      q.a2[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 32 * j * KC + H * KC));
      q.d1[j] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 32 * j * KC + 2 * H * KC));
    }
    // (rows n .. stride - 1: the planes hold the values of an all-zero input row there; the seed is zero -- they contribute nothing)
    // (guarded loads: a branch-free form with clamped indices was 20 % slower on the same box -- every lane then loads)
    q.s.x = col + 0 < a.n ? (a.seed ? a.seed[col + 0] : 1.0f) : 0.0f;
    q.s.y = col + 1 < a.n ? (a.seed ? a.seed[col + 1] : 1.0f) : 0.0f;
    q.s.z = col + 2 < a.n ? (a.seed ? a.seed[col + 2] : 1.0f) : 0.0f;
    q.s.w = col + 3 < a.n ? (a.seed ? a.seed[col + 3] : 1.0f) : 0.0f;
#pragma unroll
    for (int i = 0; i < XI; ++i) {
      const int64_t k = c * KC + xrow[i];
      q.x[i] = (xrow[i] >= 0 && k < a.n) ? a.x[k * dim + xcol[i]] : 0.0f;
    }
  };
  const auto add4 = [](f32x4 v) { return (v.x + v.y) + (v.z + v.w); };
  // the register -> LDS pass of a chunk: seed scaling, the row sums, the three planes the products read
  const auto store_chunk = [&](const Regs& q, float* st) __attribute__((always_inline)) {
    sum_seed += r0 == 0 ? add4(q.s) : 0.0f;
#pragma unroll
    for (int j = 0; j < HT; ++j) {
      // layer 2's share from its pre-activation: sigma = 1 / (1 + exp(-a2)), h2 = a2 sigma, d2 = w3 (sigma + h2 (1 - sigma)) -- the
      // arithmetic of the forward kernel's epilogue (hardware exp2 / rcp), two transcendentals per element in a memory-bound pass
      const f32x4 ex = {__builtin_amdgcn_exp2f(q.a2[j].x * -1.44269504088896340736f), __builtin_amdgcn_exp2f(q.a2[j].y * -1.44269504088896340736f),
                        __builtin_amdgcn_exp2f(q.a2[j].z * -1.44269504088896340736f), __builtin_amdgcn_exp2f(q.a2[j].w * -1.44269504088896340736f)};
      const f32x4 sg = {__builtin_amdgcn_rcpf(ex.x + 1.0f), __builtin_amdgcn_rcpf(ex.y + 1.0f), __builtin_amdgcn_rcpf(ex.z + 1.0f),
                        __builtin_amdgcn_rcpf(ex.w + 1.0f)};
      const f32x4 hu = q.a2[j] * sg;
      const f32x4 du = (sg + hu * (1.0f - sg)) * w3r[j];
      const f32x4 d2 = du * q.s, d1 = q.d1[j] * q.s, h2 = hu * q.s;
      sum_d2[j] += add4(d2);
      sum_d1[j] += add4(d1);
      sum_h2[j] += add4(h2);
      const int o = (r0 + 32 * j) * PITCH + 4 * c4;
      *reinterpret_cast<f32x4*>(st + o) = d2;
      *reinterpret_cast<f32x4*>(st + S::plane_floats + o) = q.h1[j];
      *reinterpret_cast<f32x4*>(st + 2 * S::plane_floats + o) = d1;
    }
#pragma unroll
    for (int i = 0; i < XI; ++i) st[xoff[i]] = q.x[i];
  };

  f32x16 acc2[RT][CT], acc1[DT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int j = 0; j < CT; ++j) acc2[i][j] = (f32x16)(0.0f);
#pragma unroll
  for (int j = 0; j < DT; ++j) acc1[j] = (f32x16)(0.0f);
  const int rt0 = (w >> 1) * RT, ct0 = (w & 1) * CT;
  const bool has_w1 = HT >= kThreads / 64 || w < HT;  // row tile w of dW1 (H = 128: every wave has one)

  const auto compute = [&](const float* st) __attribute__((always_inline)) {
    const float* d2s = st;
    const float* h1s = st + S::plane_floats;
    const float* d1s = st + 2 * S::plane_floats;
    const float* xs = st + 3 * S::plane_floats;
#pragma unroll
    for (int tt = 0; tt < KC / 8; ++tt) {
      const int ko = 8 * tt + 4 * hf;
      f32x4 av[RT], bv[CT];
#pragma unroll
      for (int i = 0; i < RT; ++i) av[i] = *reinterpret_cast<const f32x4*>(d2s + (32 * (rt0 + i) + r) * PITCH + ko);
#pragma unroll
      for (int j = 0; j < CT; ++j) bv[j] = *reinterpret_cast<const f32x4*>(h1s + (32 * (ct0 + j) + r) * PITCH + ko);
      f32x4 a1 = (f32x4)(0.0f);
      float xb[4][DT];
      if (has_w1) {
        a1 = *reinterpret_cast<const f32x4*>(d1s + (32 * w + r) * PITCH + ko);
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
            acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][e], bv[j][e], acc2[i][j], 0, 0, 0);
        if (has_w1) {
#pragma unroll
          for (int j = 0; j < DT; ++j) acc1[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[e], xb[e][j], acc1[j], 0, 0, 0);
        }
      }
    }
  };

  // A workgroup's chunk k = tile blockIdx.x + k G.  ONE LDS stage and one register set per workgroup, TWO workgroups per CU: while
  // one multiplies a chunk (80 MFMAs of 64 cycles per wave) the other runs its register -> LDS pass or waits for its loads.  Measured
  // (MI355X, n = 131 072) on the two-stage / one-workgroup form this replaces: 83 us = 27 (loads alone: the planes mostly hit the
  // memory-side cache) + 15 (the pass) + 41 (the products) -- the three phases of a lone wave per SIMD add up whatever their order in
  // the program (one or two chunks in flight, the pass behind or between the MFMAs: the same 83 - 87 us).  A split into ROLES -- 512
  // threads, on every SIMD one wave that only multiplies and one that only loads and stages, two LDS stages -- measured 82 us (+ 8 for 256
  // records) against this form's 74 (+ 14) on the same box.  (Round 5 read that as "a SIMD does not run one wave's vector work under another
  // wave's MFMAs"; the pinned-stream measurement of round 6, profiles/r06_mfma_valu_overlap.txt, says it does -- at ~one vector instruction
  // per 5 cycles in total beside a busy pipe -- and this launch is bound by its memory traffic, not by either.)
  const int64_t nk = (a.chunks - blockIdx.x + G - 1) / G;
  const auto chunk_of = [&](int64_t k) { return (int64_t)blockIdx.x + k * G; };
  load_chunk(rq, chunk_of(0));
  __syncthreads();  // the zeroed x columns
  store_chunk(rq, stage);
  __syncthreads();
  for (int64_t k = 0; k < nk; ++k) {
    if (k + 1 < nk) load_chunk(rq, chunk_of(k + 1));  // in flight behind the products
    __builtin_amdgcn_sched_barrier(0);
    compute(stage);
    __builtin_amdgcn_sched_barrier(0);
    if (k + 1 >= nk) break;
    __syncthreads();  // every wave is done reading chunk k
    store_chunk(rq, stage);
    __syncthreads();
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
  if (e < S::record) {
#pragma unroll 8  // (eight loads in flight per thread; the sum keeps its order)
    for (int g = q; g < G; g += 8) s += __builtin_nontemporal_load(&partials[(int64_t)g * S::record + e]);
  }
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
  cus *= 2;  // two workgroups per CU (__launch_bounds__(256, 2), one 60 KB stage each)
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

// floats of workspace ebm_mlp_param_grads_f32 needs on the current device (one partial record per workgroup, two workgroups per CU)
int64_t mlp_param_grads_work_floats(int32_t hidden, int32_t dim, int64_t n) {
  const int64_t chunks = (n + 127) / 128 * 128 / mlpgrads::KC;
  const int64_t dp = 32 * ((dim + 31) / 32);
  const int64_t record = (int64_t)hidden * dp + hidden + (int64_t)hidden * hidden + 2 * hidden + 1;
  return record * mlpgrads::grid_of(chunks > 0 ? chunks : 1);
}

int launch_mlp_param_grads(int32_t hidden, const float* acts, const float* x, int64_t n, int32_t dim, const float* seed, const float* w3,
                           float* work, float* out, hipStream_t st, const char* who) {
  mlpgrads::Args a{};
  a.acts = acts; a.w3 = w3; a.stride = (n + 127) / 128 * 128; a.x = x; a.n = n; a.dim = dim; a.seed = seed; a.partials = work;
  a.chunks = a.stride / mlpgrads::KC;
  const int dt = (dim + 31) / 32;
  if (hidden == 64) return dt == 1 ? mlpgrads::launch<2, 1>(a, out, st, who) : mlpgrads::launch<2, 2>(a, out, st, who);
  return dt == 1 ? mlpgrads::launch<4, 1>(a, out, st, who) : mlpgrads::launch<4, 2>(a, out, st, who);
}

}  // namespace ebm
