// The wide-MLP Langevin chain at TWO waves per SIMD (round 5): H = 128, dim <= 32, the plain call -- config 5's own sampler
// call (65 536 chains x 2, k = 20) and the reference's benchmark network at dim 8 / 32 (benchmarks/registry.py:372-387).
//
// Why.  mlp_wide_chain_kernel<4, 1, 2, 1> gives a wave a whole 32-chain tile: four output tiles per contraction, the h1 / d2
// operands of all 128 hidden units in registers -- 371 registers, ONE wave per SIMD, and a lone wave issues one instruction per
// 6 - 7 cycles whatever it is (profiles/r03_pmc_mlp.txt): 4 300 instructions per step next to 480 MFMAs of 32 cycles leave the
// bf16 pipe 48 - 58 % busy.  A second wave per SIMD that carries its own chain tile needs the same ~370 registers; it was tried
// in round 3 (docs/design/mlp_wide.md: 226 spills, 1.6x slower).
//
// What.  FOUR waves share one 32-chain tile and each owns ONE of the four hidden tiles -- rows 32 q .. 32 q + 31 of both hidden
// layers, through the whole evaluation:
//   a1_q = W1[q rows] x + b1        h1_q = silu(a1_q), s1_q = silu'(a1_q)          (needs all of x:   x exchange, 6 KB)
//   a2_q = W2[q rows] h1 + b2       d2_q = w3 silu'(a2_q)                          (needs all of h1:  operand exchange, 24 KB)
//   T_q  = W2^T[q rows] d2          d1_q = T_q s1_q                                (needs all of d2:  the same 24 KB)
//   g   += W1^T[., q rows] d1_q                                                    (K = the wave's OWN rows: no exchange of d1;
//                                                                                    the four partial g tiles are summed through LDS)
// A wave's live set is one accumulator pair, s1 and a register quad of the state: ~130 registers, so a 512-thread workgroup =
// two chain tiles = two waves per SIMD fits next to the 96 KB image of W2 (one copy for both tiles).  What crosses waves is the
// B operand of the next contraction, already split three ways: a lane writes the 16 bytes (K-block, piece) that the SAME lane of
// the other three waves reads -- the C/D layout never changes lanes (mlp_wide.hip), so the exchange is [K-block][piece][lane]
// with one ds_write_b128 / ds_read_b128 per slot, conflict-free by construction.  W1 never enters LDS: the two operand sets a
// wave needs of it (its 32 rows for W1 x, the same rows transposed for W1^T d1) are 12 - 48 registers, split once per launch.
//
// Phases.  One evaluation + update is six phases per wave, separated by workgroup barriers because every phase reads what
// the previous one wrote to the tile's single exchange buffer:
//   0  read x (B operand), W1 x on the matrix pipe, SiLU / SiLU', split h1, WRITE h1           VALU-dense
//   1  W2 h1: 48 MFMAs, A from the image, B from the exchange                                    matrix-dense
//   2  SiLU' of a2, d2 = w3 silu'(a2), split, WRITE d2                                           VALU-dense
//   3  W2^T d2: 48 MFMAs (transposed reads of the same image)                                    matrix-dense
//   4  d1 = T s1, split, W1^T d1 over the wave's own K (12 MFMAs), WRITE the partial gradient     mixed
//   5  read the four partials of the wave's register quad, Euler-Maruyama update with in-kernel Philox draws, split, WRITE x
// The two chain tiles of a workgroup run the same sequence THREE PHASES APART (the tick loop below): whenever one tile's waves
// are in a matrix-dense phase, the partner wave on the same SIMD is in a VALU-dense one -- the two pipes of a SIMD are separate
// (MI355X_MICROARCH.md, "Two waves per SIMD"), and this pairing is what keeps both busy; run in lockstep (QOFF = 0) the same
// code leaves the matrix pipe idle during every epilogue.
//
// Arithmetic: the same six bf16 products per K-block in the same order as mlp_b16.h (lo | mid | hi of the weight, smallest
// first; even terms into one accumulator, odd into a second -- contract_pipe's NT == 1 form), K-blocks 0 .. 7 in order; the
// gradient is the sum of the four partial tiles in wave order.  Same draws as every other route: the field is addressed by
// (seed, step, flat element) (ebm_common.h).  Reference: torchebm/samplers/langevin_dynamics.py:154-185,
// core/base_integrator.py:711-731, examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-43 (the network).
#include "mlp_wide_body.h"

namespace ebm {
namespace widemlp {

using namespace mlpb16;

constexpr int kQuadBlock = 512;
constexpr int kQH = 128;
// LDS carve-up (bytes): [b1 b2 w3][x exchange: 2 tiles][operand exchange: 2 tiles][W2 image]
constexpr uint32_t kQBias = 0u, kQBiasBytes = 3u * kQH * 4u;                 // 1 536
constexpr uint32_t kQXe = kQBias + kQBiasBytes, kQXeTile = 2u * 3u * 1024u;  // 6 144 per tile
constexpr uint32_t kQX = kQXe + 2u * kQXeTile, kQXTile = 8u * 3u * 1024u;    // 24 576 per tile
constexpr uint32_t kQImg = kQX + 2u * kQXTile;                               // 62 976
constexpr uint32_t kQFlag = kQImg + 3u * kQH * kQH * 2u;                     // 161 280: the two tile-barrier counters
constexpr uint32_t kQSmem = kQFlag + 16u;                                    // 161 296 of 163 840
constexpr uint32_t kQSplit = (uint32_t)kQH * 2u * kQH;                       // 32 768: one split of the image

typedef __attribute__((address_space(3))) u32x4* lds_u32x4;
typedef __attribute__((address_space(3))) u32x2* lds_u32x2;
typedef __attribute__((address_space(3))) f32x4* lds_f32x4;

__device__ __forceinline__ u32x4 lds_load16(uint32_t addr) { return *(lds_u32x4)(uintptr_t)addr; }
__device__ __forceinline__ void lds_store16(uint32_t addr, u32x4 v) { *(lds_u32x4)(uintptr_t)addr = v; }
__device__ __forceinline__ bf16x8 as_bf16(u32x4 v) { return __builtin_bit_cast(bf16x8, v); }

// the six products of one K-block: a = the weight's pieces, b = the operand's; even terms -> acc0, odd -> acc1
__device__ __forceinline__ void quad_terms(f32x16& acc0, f32x16& acc1, const u32x4& al, const u32x4& am, const u32x4& ah,
                                           const Split8p& b) {
  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(al), as_bf16(b.h), acc0, 0, 0, 0);  // 0: Al dh
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(am), as_bf16(b.m), acc1, 0, 0, 0);  // 1: Am dm
  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(am), as_bf16(b.h), acc0, 0, 0, 0);  // 2: Am dh
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(ah), as_bf16(b.l), acc1, 0, 0, 0);  // 3: Ah dl
  acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(ah), as_bf16(b.m), acc0, 0, 0, 0);  // 4: Ah dm
  acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(ah), as_bf16(b.h), acc1, 0, 0, 0);  // 5: Ah dh
}

// eight fp32 values (one K-block of one lane) -> the three packed pieces
__device__ __forceinline__ void split8(const float (&v)[8], Split8p& s) {
  static_for<4>([&](auto pc) __attribute__((always_inline)) {
    constexpr int p = decltype(pc)::value;
    f32x2 r;
    pair_split_a<p>(s, r, (f32x2){v[2 * p], v[2 * p + 1]});
    pair_split_b<p>(s, r);
    pair_split_c<p>(s, r);
  });
}

// -DEBM_PHASE_TIMES (scripts/quad_phase_times.py only): waves 0 and 4 of workgroup 0 (the two chain tiles' first waves, one SIMD)
// log the shader clock at the start of a tick's phase, at its end and behind the barrier: [wave q0 of tile g][tick][3].
#ifdef EBM_PHASE_TIMES
__device__ unsigned long long ebm_quad_log[2 * 512 * 3];
#define EBM_QSTAMP(WHICH)                                                                                        \
  do {                                                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
    if (__builtin_amdgcn_readfirstlane(blockIdx.x) == 0 && q == 0 && tick < 512) {                                \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                              \
      if (lane == 0) ebm_quad_log[(grp * 512 + tick) * 3 + (WHICH)] = now_;                                       \
    }                                                                                                            \
    __builtin_amdgcn_sched_barrier(0);                                                                           \
  } while (0)
#else
#define EBM_QSTAMP(WHICH) do {} while (0)
#endif

// KB1: K-blocks of the input width (1: dim <= 16, 2: dim <= 32).  QOFF: phases between the two chain tiles of a workgroup.
template <int KB1, int QOFF>
__global__ __launch_bounds__(kQuadBlock, 2) void mlp_quad_chain_kernel(WideArgs a) {
  constexpr int H = kQH;
  const uint32_t smem = (uint32_t)(uintptr_t)(lds_bytes)wide_smem;
  const int dim = a.dim;
  const float* W1g = a.params;
  const float* b1g = W1g + H * dim;
  const float* W2g = b1g + H;
  {
    const float* b2g = W2g + H * H;
    const float* w3g = b2g + H;
    stage_image<H, H>(W2g, H, H, (lds_bytes)wide_smem + kQImg, kQuadBlock);
    if (threadIdx.x < 4) *(__attribute__((address_space(3))) uint32_t*)(uintptr_t)(smem + kQFlag + 4u * threadIdx.x) = 0u;
    for (int i = threadIdx.x; i < H; i += kQuadBlock) {
      wide_smem[i] = b1g[i];
      wide_smem[H + i] = b2g[i];
      wide_smem[2 * H + i] = w3g[i];
    }
  }
  const int lane = threadIdx.x & 63, m = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wave >> 2, q = wave & 3;  // chain tile of the workgroup | hidden tile (and register quad of the state) owned
  const int64_t sample = ((int64_t)blockIdx.x * 2 + grp) * 32 + m;
  const bool active = sample < a.n_chains;
  const bool quads = (dim & 3) == 0;

  // ---- W1 as A operands, in registers for the whole launch
  //   forward (W1 x):        lane (m, h), K-block kb, element j: W1[32 q + m][16 kb + 8 (j >> 2) + 4 h + (j & 3)]
  //   transposed (W1^T d1):  lane (m, h), K-block kl of the wave's own 32 rows, element j: W1[32 q + 16 kl + 8 (j >> 2) + 4 h + (j & 3)][m]
  Split8p w1f[KB1], w1t[2];
  static_for<KB1>([&](auto kbc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kbc)::value;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 16 * kb + 8 * (j >> 2) + 4 * h + (j & 3);
      v[j] = c < dim ? W1g[(32 * q + m) * dim + c] : 0.0f;
    }
    split8(v, w1f[kb]);
  });
  static_for<2>([&](auto klc) __attribute__((always_inline)) {
    constexpr int kl = decltype(klc)::value;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int row = 32 * q + 16 * kl + 8 * (j >> 2) + 4 * h + (j & 3);
      v[j] = m < dim ? W1g[row * dim + m] : 0.0f;
    }
    split8(v, w1t[kl]);
  });

  // ---- the wave's register quad of the state: columns c0 .. c0 + 3 of chain m
  const int c0 = 8 * q + 4 * h;
  float xq[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) xq[i] = (active && c0 + i < dim) ? a.x[sample * dim + c0 + i] : 0.0f;

  // ---- LDS addresses
  const uint32_t xe = smem + kQXe + (uint32_t)grp * kQXeTile + (uint32_t)lane * 16u;  // x exchange [kb][piece][lane]
  const uint32_t xb = smem + kQX + (uint32_t)grp * kQXTile + (uint32_t)lane * 16u;    // operand exchange [kb][piece][lane]
  const uint32_t bias_q = smem + kQBias + (uint32_t)(32 * q + 4 * h) * 4u;             // + 512 layer, + 32 per register quad
  // forward walk over the image: row 32 q + m, unit h (swizzled), K-block kb XORs 32 kb in
  const uint32_t fw_row = smem + kQImg + (uint32_t)(32 * q + m) * Img<H>::RB;
  const uint32_t fw_x = (16u * (uint32_t)h) ^ Img<H>::swz((uint32_t)m);
  // transposed walk: (tile, half) registers as in Walk::bwd_tile, the tile being the wave's
  uint32_t tw0, tw1;
  {
    const uint32_t i = lane & 15, g = (lane >> 4) & 1, row = 4u * (uint32_t)h + (i >> 2);
    const uint32_t bx = (32u * g + 16u * (i & 1u) + 8u * ((i >> 1) & 1u)) ^ Img<H>::swz(row);
    tw0 = smem + kQImg + row * Img<H>::RB + (bx ^ (64u * (uint32_t)q));
    tw1 = smem + kQImg + row * Img<H>::RB + (bx ^ (64u * (uint32_t)q) ^ Img<H>::swz(8u)) + 8u * Img<H>::RB;
  }

  const auto split_x_write = [&]() __attribute__((always_inline)) {  // the quad as half a K-block of the x exchange
    if (q < 2 * KB1) {
      Split8p s;
      static_for<2>([&](auto pc) __attribute__((always_inline)) {
        constexpr int p = decltype(pc)::value;
        f32x2 r;
        pair_split_a<p>(s, r, (f32x2){xq[2 * p], xq[2 * p + 1]});
        pair_split_b<p>(s, r);
        pair_split_c<p>(s, r);
      });
      const uint32_t dst = xe + (uint32_t)(q >> 1) * 3072u + 8u * (uint32_t)(q & 1);
      *(lds_u32x2)(uintptr_t)(dst) = (u32x2){s.h[0], s.h[1]};
      *(lds_u32x2)(uintptr_t)(dst + 1024u) = (u32x2){s.m[0], s.m[1]};
      *(lds_u32x2)(uintptr_t)(dst + 2048u) = (u32x2){s.l[0], s.l[1]};
    }
  };
  split_x_write();
  __syncthreads();  // image, biases and the first x are in place

  f32x16 acc;        // a2 (phase 1 -> 2) | T (phase 3 -> 4)
  f32x2 s1[8];       // silu'(a1) of the wave's rows (phase 0 -> 4)
  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;
  int until_keep = a.thin;
  int64_t keep_off = 0;

  const auto zero16 = [](f32x16& t) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 16; ++r) t[r] = 0.0f;
  };
  const auto bias16 = [&](f32x16& t, uint32_t layer_off) __attribute__((always_inline)) {
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const f32x4 bq = *(lds_f32x4)(uintptr_t)(bias_q + layer_off + 32u * (uint32_t)qq);
      t[4 * qq] = bq[0]; t[4 * qq + 1] = bq[1]; t[4 * qq + 2] = bq[2]; t[4 * qq + 3] = bq[3];
    }
  };
  // The tile epilogues, STAGE-major over the eight register pairs of the tile: a wave that shares its SIMD with one other wave
  // hides no latency by itself, and a pair's epilogue is a chain of ~20 dependent instructions (packed multiply, exp, add, rcp
  // ... three conversions with their residuals); pair-major -- what the scheduler makes of the obvious loop: two pairs in
  // flight -- a lone wave spent 9.5 cycles per instruction on it (scripts/quad_phase_times.py).  Stage-major every instruction
  // has seven independent neighbours between itself and its consumer.  The fences keep the stages apart.
#define EBM_QFENCE() __builtin_amdgcn_sched_barrier(0)
  // sg = sigmoid(a), hv = a sg (= silu), sp = silu'(a) = sg + hv (1 - sg), for the eight pairs of `t`
  const auto silu_stages = [&](const f32x16& t, f32x2 (&hv)[8], f32x2 (&sp)[8]) __attribute__((always_inline)) {
    f32x2 ee[8], sg[8];
    static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; ee[i] = pair_of<2 * i>(t) * -1.44269504088896340736f; });
    EBM_QFENCE();
    static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; ee[i] = (f32x2){__builtin_amdgcn_exp2f(ee[i].x), __builtin_amdgcn_exp2f(ee[i].y)}; });
    EBM_QFENCE();
    static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; ee[i] = ee[i] + 1.0f; });
    EBM_QFENCE();
    static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; sg[i] = (f32x2){__builtin_amdgcn_rcpf(ee[i].x), __builtin_amdgcn_rcpf(ee[i].y)}; });
    EBM_QFENCE();
    static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; hv[i] = pair_of<2 * i>(t) * sg[i]; ee[i] = 1.0f - sg[i]; });
    EBM_QFENCE();
    static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; sp[i] = __builtin_elementwise_fma(hv[i], ee[i], sg[i]); });
    EBM_QFENCE();
  };
  // the eight pairs -> the three pieces of the tile's two K-blocks
  const auto split_stages = [&](const f32x2 (&v)[8], Split8p (&s)[2]) __attribute__((always_inline)) {
    f32x2 rs[8];
    static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; pair_split_a<(i & 3)>(s[i >> 2], rs[i], v[i]); });
    EBM_QFENCE();
    static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; pair_split_b<(i & 3)>(s[i >> 2], rs[i]); });
    EBM_QFENCE();
    static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; pair_split_c<(i & 3)>(s[i >> 2], rs[i]); });
    EBM_QFENCE();
  };
  // the wave's two K-blocks (its 16 registers, by pairs) -> slots 2 q, 2 q + 1 of the operand exchange
  const auto write_own = [&](const Split8p (&s)[2]) __attribute__((always_inline)) {
    const uint32_t dst = xb + (uint32_t)q * 6144u;
    lds_store16(dst, s[0].h); lds_store16(dst + 1024u, s[0].m); lds_store16(dst + 2048u, s[0].l);
    lds_store16(dst + 3072u, s[1].h); lds_store16(dst + 4096u, s[1].m); lds_store16(dst + 5120u, s[1].l);
  };
  // 48 MFMAs over the eight K-blocks of the exchange; TR: the transposed walk over the image.  A K-block's operands are
  // requested TWO K-blocks ahead, one or two requests behind each MFMA of the block in between (in a block of their own in
  // front of the MFMAs they cost the matrix pipe ~50 idle cycles per K-block: a lone wave took 2 400 cycles for these 1 536);
  // fill(slot), slot = 0 .. 47: the caller's independent work behind MFMA `slot`.
  const auto contract8 = [&](f32x16& acc0, auto trc, auto fill) __attribute__((always_inline)) {
    constexpr bool TR = decltype(trc)::value;
    f32x16 acc1;
    zero16(acc1);
    u32x4 pa[3][3];
    u32x4 pbv[3][3];
    uint32_t fw = fw_row, t0 = tw0, t1 = tw1, xbo = xb;
    asm volatile("" : "+v"(fw), "+v"(t0), "+v"(t1), "+v"(xbo));  // (the image never changes: keep its loads at their MFMAs)
    uint32_t t0f = t0 + 65536u, t1f = t1 + 65536u;
    // request r = 0 .. 5 of K-block kb into buffer buf: r < 3: piece r of A (0: hi), r >= 3: piece r - 3 of B
    const auto request = [&](auto kbc, auto bufc, auto rc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value, buf = decltype(bufc)::value, r = decltype(rc)::value;
      if constexpr (r < 3) {
        if constexpr (TR) {
          constexpr uint32_t off = (uint32_t)r * kQSplit + 16u * kb * Img<H>::RB;
          bf16x4 lo4, hi4;
          if constexpr (off < 65536u) {
            lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(uintptr_t)(t0 + off));
            hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(uintptr_t)(t1 + off));
          } else {
            lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(uintptr_t)(t0f + (off - 65536u)));
            hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)(uintptr_t)(t1f + (off - 65536u)));
          }
          pa[buf][r] = __builtin_bit_cast(u32x4, __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7));
        } else {
          const uint32_t ad = fw + (fw_x ^ (32u * kb));
          pa[buf][r] = lds_load16(ad + (uint32_t)r * kQSplit);
        }
      } else {
        pbv[buf][r - 3] = lds_load16(xbo + (3u * kb + (uint32_t)(r - 3)) * 1024u);
      }
    };
    static_for<2>([&](auto kbc) __attribute__((always_inline)) {
      static_for<6>([&](auto rc) __attribute__((always_inline)) { request(kbc, kbc, rc); });
    });
    EBM_QFENCE();
    static_for<8>([&](auto kbc) __attribute__((always_inline)) {
      constexpr int kb = decltype(kbc)::value, buf = kb % 3, nbuf = (kb + 2) % 3;
      static_for<6>([&](auto tc) __attribute__((always_inline)) {
        constexpr int term = decltype(tc)::value;  // 0: Al dh | 1: Am dm, 2: Am dh | 3: Ah dl, 4: Ah dm, 5: Ah dh
        constexpr int as = term == 0 ? 2 : (term < 3 ? 1 : 0);
        constexpr int bs = (term == 0 || term == 2 || term == 5) ? 0 : ((term == 1 || term == 4) ? 1 : 2);
        if constexpr (term & 1) acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(pa[buf][as]), as_bf16(pbv[buf][bs]), acc1, 0, 0, 0);
        else acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf16(pa[buf][as]), as_bf16(pbv[buf][bs]), acc0, 0, 0, 0);
        if constexpr (kb + 2 < 8) request(std::integral_constant<int, kb + 2>{}, std::integral_constant<int, nbuf>{}, tc);
        fill(std::integral_constant<int, 6 * kb + term>{});
        EBM_QFENCE();
      });
    });
    acc0 += acc1;
  };
  const auto no_fill = [](auto) __attribute__((always_inline)) {};

  // The step's draws, cut into slots that run behind the MFMAs of phase 3 (the wave is matrix-bound there and the update of
  // phase 5 would otherwise wait out ten Philox rounds and both Box-Muller chains with nothing to overlap): the same
  // arithmetic as normal4_at (ebm_common.h), instruction for instruction.
  uint32_t pc0 = 0, pc1 = 0, pc2 = 0, pc3 = 0, pk0 = 0, pk1 = 0;
  float bm_a = 0.0f, bm_b = 0.0f, bm_ra = 0.0f, bm_rb = 0.0f, bm_sa = 0.0f, bm_ca = 0.0f, bm_sb = 0.0f, bm_cb = 0.0f;
  float eps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const auto draw_slot = [&](auto sc, int step_now) __attribute__((always_inline)) {
    constexpr int sl = decltype(sc)::value;
    if constexpr (sl == 0) {
      int64_t smp = sample;
      asm volatile("" : "+v"(smp));
      const uint64_t group = quads ? (((uint64_t)smp * (uint64_t)dim + (uint64_t)c0) >> 2) : ((uint64_t)smp >> 1);
      const uint64_t st = a.step0 + (uint64_t)step_now;
      pc0 = (uint32_t)group; pc1 = (uint32_t)(group >> 32); pc2 = (uint32_t)st; pc3 = (uint32_t)(st >> 32);
      pk0 = a.key.k0; pk1 = a.key.k1;
    } else if constexpr (sl <= 10) {  // one Philox round (philox4x32_10)
      const uint64_t p0 = (uint64_t)0xD2511F53u * pc0;
      const uint64_t p1 = (uint64_t)0xCD9E8D57u * pc2;
      const uint32_t n0 = xor3((uint32_t)(p1 >> 32), pc1, pk0);
      const uint32_t n2 = xor3((uint32_t)(p0 >> 32), pc3, pk1);
      pc1 = (uint32_t)p1; pc3 = (uint32_t)p0; pc0 = n0; pc2 = n2;
      pk0 += 0x9E3779B9u; pk1 += 0xBB67AE85u;
    } else if constexpr (sl == 11) {  // box_muller: u1, rev
      bm_a = __builtin_amdgcn_logf(u01_open_low(pc0));
      bm_b = __builtin_amdgcn_logf(u01_open_low(pc2));
    } else if constexpr (sl == 12) {
      bm_ra = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * bm_a);
      bm_rb = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * bm_b);
    } else if constexpr (sl == 13) {
      const float reva = (float)pc1 * 0x1p-32f;
      bm_sa = __builtin_amdgcn_sinf(reva); bm_ca = __builtin_amdgcn_cosf(reva);
    } else if constexpr (sl == 14) {
      const float revb = (float)pc3 * 0x1p-32f;
      bm_sb = __builtin_amdgcn_sinf(revb); bm_cb = __builtin_amdgcn_cosf(revb);
    } else if constexpr (sl == 15) {
      const float n0 = bm_ra * bm_sa, n1 = bm_ra * bm_ca, n2 = bm_rb * bm_sb, n3 = bm_rb * bm_cb;
      if (quads) {  // the quad is exactly one Philox counter
        eps[0] = n0; eps[1] = n1; eps[2] = n2; eps[3] = n3;
      } else {  // dim == 2 (config 5's shape): a chain's two elements are half a Philox counter
        const bool odd = (sample & 1) != 0;
        eps[0] = odd ? n2 : n0; eps[1] = odd ? n3 : n1; eps[2] = 0.0f; eps[3] = 0.0f;
      }
      if (c0 >= dim) { eps[0] = 0.0f; eps[1] = 0.0f; eps[2] = 0.0f; eps[3] = 0.0f; }
    }
  };

  // ---- a barrier of ONE chain tile (its four waves): an LDS counter every wave bumps on arrival and polls until the whole
  // tile has.  The workgroup's other tile is not involved: the two tiles run free of each other, and while the waves of one
  // wait here (or on any latency) the partner wave of the same SIMD -- the other tile's -- has the issue slots.  LDS
  // operations of a wave are carried out in order, so the bump follows the wave's own reads / writes of the exchange, and
  // what a wave reads after it saw the full count follows every other wave's.  The poll is bounded (a miscounted barrier
  // gives wrong samples, never a hung GPU).
  uint32_t epoch = 0;
  const uint32_t flag = smem + kQFlag + 4u * (uint32_t)grp;
  const auto tile_sync = [&]() __attribute__((always_inline)) {
    epoch += 4u;
    if (lane == 0) {
      const uint32_t one = 1u;
      asm volatile("ds_add_u32 %0, %1" ::"v"(flag), "v"(one) : "memory");
    }
    for (int spin = 0; spin < (1 << 20); ++spin) {
      uint32_t seen;
      asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(seen) : "v"(flag) : "memory");
      if (__builtin_amdgcn_readfirstlane(seen) >= epoch) break;
      __builtin_amdgcn_s_sleep(1);
    }
  };
#ifdef EBM_QUAD_SOLO  // (phase-time experiments: tile 1 idles)
  const int n_steps_here = grp == 0 ? a.k_steps : 0;
#else
  const int n_steps_here = a.k_steps;
#endif
  for (int step = 0; step < n_steps_here; ++step) {
    [[maybe_unused]] int tick = 6 * step;
    {
      EBM_QSTAMP(0);
      {
        // ---------------------------------------------------------------- layer 1 + epilogue 1
        f32x16 u0, u1;
        bias16(u0, 0u);
        zero16(u1);
        uint32_t xeo = xe;
        asm volatile("" : "+v"(xeo));
        static_for<KB1>([&](auto kbc) __attribute__((always_inline)) {
          constexpr int kb = decltype(kbc)::value;
          Split8p b;
          b.h = lds_load16(xeo + (3u * kb) * 1024u);
          b.m = lds_load16(xeo + (3u * kb + 1u) * 1024u);
          b.l = lds_load16(xeo + (3u * kb + 2u) * 1024u);
          quad_terms(u0, u1, w1f[kb].l, w1f[kb].m, w1f[kb].h, b);
        });
        u0 += u1;
        EBM_QFENCE();
        f32x2 hv[8];
        silu_stages(u0, hv, s1);
        Split8p sph[2];
        split_stages(hv, sph);
        write_own(sph);
      }
      EBM_QSTAMP(1);
      tile_sync();
      EBM_QSTAMP(2);
      ++tick;
      EBM_QSTAMP(0);
      {
        // ---------------------------------------------------------------- layer 2: a2 = W2[q rows] h1 + b2
        bias16(acc, 512u);
        contract8(acc, std::false_type{}, no_fill);
      }
      EBM_QSTAMP(1);
      tile_sync();
      EBM_QSTAMP(2);
      ++tick;
      EBM_QSTAMP(0);
      {
        // ---------------------------------------------------------------- epilogue 2: d2 = w3 silu'(a2)
        f32x2 hv[8], sp[8];
        f32x4 w3q[4];
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) w3q[qq] = *(lds_f32x4)(uintptr_t)(bias_q + 1024u + 32u * (uint32_t)qq);
        silu_stages(acc, hv, sp);
        static_for<8>([&](auto ic) __attribute__((always_inline)) {
          constexpr int i = decltype(ic)::value;
          const f32x2 w3p = (i & 1) ? (f32x2){w3q[i >> 1][2], w3q[i >> 1][3]} : (f32x2){w3q[i >> 1][0], w3q[i >> 1][1]};
          sp[i] = w3p * sp[i];
        });
        EBM_QFENCE();
        Split8p spd[2];
        split_stages(sp, spd);
        write_own(spd);
      }
      EBM_QSTAMP(1);
      tile_sync();
      EBM_QSTAMP(2);
      ++tick;
      EBM_QSTAMP(0);
      {
        // ---------------------------------------------------------------- backward through W2: T = W2^T[q rows] d2
        zero16(acc);
                contract8(acc, std::true_type{}, [&](auto oc) __attribute__((always_inline)) {
          constexpr int o = decltype(oc)::value;  // the draws of this step behind the first MFMAs (two slots of matrix time each)
          if constexpr ((o & 1) == 0 && o / 2 <= 15) draw_slot(std::integral_constant<int, o / 2>{}, step);
        });
      }
      EBM_QSTAMP(1);
      tile_sync();
      EBM_QSTAMP(2);
      ++tick;
      EBM_QSTAMP(0);
      {
        // ---------------------------------------------------------------- d1 = T s1; the wave's share of W1^T d1
        f32x2 dd[8];
        static_for<8>([&](auto ic) __attribute__((always_inline)) { constexpr int i = decltype(ic)::value; dd[i] = pair_of<2 * i>(acc) * s1[i]; });
        EBM_QFENCE();
        Split8p spe[2];
        split_stages(dd, spe);
        f32x16 g0, g1;
        zero16(g0);
        zero16(g1);
        quad_terms(g0, g1, w1t[0].l, w1t[0].m, w1t[0].h, spe[0]);
        quad_terms(g0, g1, w1t[1].l, w1t[1].m, w1t[1].h, spe[1]);
        g0 += g1;
        // partial gradient: register quad j goes to the wave that owns it -- [quad j][from wave q][lane]
        const uint32_t dst = xb + (uint32_t)q * 1024u;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          *(lds_f32x4)(uintptr_t)(dst + 4096u * (uint32_t)j) = (f32x4){g0[4 * j], g0[4 * j + 1], g0[4 * j + 2], g0[4 * j + 3]};
      }
      EBM_QSTAMP(1);
      tile_sync();
      EBM_QSTAMP(2);
      ++tick;
      EBM_QSTAMP(0);
      {
        // ---------------------------------------------------------------- Euler-Maruyama update of the wave's quad (reference op order)
        const uint32_t src = xb + (uint32_t)q * 4096u;
        const f32x4 p0 = *(lds_f32x4)(uintptr_t)(src), p1 = *(lds_f32x4)(uintptr_t)(src + 1024u);
        const f32x4 p2 = *(lds_f32x4)(uintptr_t)(src + 2048u), p3 = *(lds_f32x4)(uintptr_t)(src + 3072u);
        const f32x4 gq = ((p0 + p1) + p2) + p3;
        if (a.table) {
          const float4 tb = a.table[step];
          eta = tb.x; sqrt_eta = tb.y; noise_coef = tb.z;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x1 = xq[i] - eta * gq[i];
          const float dw = eps[i] * sqrt_eta;
          // no select: a padding column has x = 0, g = 0 exactly (its W1 operand is zero) and no draw -- it stays 0 by itself
          xq[i] = x1 + noise_coef * dw;
        }
        if (a.traj && --until_keep == 0) {
          until_keep = a.thin;
          if (active) {
            float* dstp = a.traj + sample * (int64_t)a.n_kept * dim + keep_off;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (c0 + i < dim) dstp[c0 + i] = xq[i];
          }
          keep_off += dim;
        }
        split_x_write();
      }
      EBM_QSTAMP(1);
      tile_sync();
      EBM_QSTAMP(2);
    }
  }
  if (active) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (c0 + i < dim) a.x[sample * dim + c0 + i] = xq[i];
  }
}

#ifndef EBM_QOFF
#define EBM_QOFF 3
#endif
template <int KB1>
static int launch_quad_kb(const WideArgs& a, hipStream_t st, const char* who) {
  static DeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_quad_chain_kernel<KB1, EBM_QOFF>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)kQSmem);
  const int64_t blocks = ceil_div64(a.n_chains, 64);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "%s: too many chains for one launch", who);
  hipLaunchKernelGGL((mlp_quad_chain_kernel<KB1, EBM_QOFF>), dim3((unsigned)blocks), dim3(kQuadBlock), kQSmem, st, a);
  return check_launch(who);
}

// H = 128, dim <= 32, the plain call (wide_fast_shape, no records)
int launch_quad(const WideArgs& a, hipStream_t st, const char* who) {
  return a.dim <= 16 ? launch_quad_kb<1>(a, st, who) : launch_quad_kb<2>(a, st, who);
}

}  // namespace widemlp
}  // namespace ebm

#ifdef EBM_PHASE_TIMES
extern "C" __attribute__((visibility("default"))) int ebm_debug_quad_log(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ebm::widemlp::ebm_quad_log), (size_t)n * sizeof(unsigned long long));
}
#endif
