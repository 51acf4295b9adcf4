// (kernels and launch templates; instantiated by gauss_big.hip -- tiled kernels, public entry points -- and gauss_res.hip -- register-resident kernels)
// k-fused Langevin chain for the dense Gaussian energy at dims 132 .. 512 (multiples of 4) on the bf16 matrix pipe.
//
//   reference: torchebm/samplers/langevin_dynamics.py:150-185 (the step loop), torchebm/core/base_model.py (GaussianModel:
//   E = 0.5 (x - mu)^T P (x - mu), gradient P (x - mu))
//
// Below 129 the chain state lives in registers for the whole call and the three bf16 splits of Ps are resident in LDS
// (gauss_mfma.hip).  Here the splits of Ps are 0.4 - 1.5 MB: Ps STREAMS -- per stage of two K-blocks the workgroup loads the
// [32 OT] x 32 slab of Ps (fp32, symmetric), splits it into three bf16 pieces and writes them operand-ready to LDS
// (double-buffered, one barrier per stage); every wave reads each image with one ds_read_b128 per (tile, K-block).  Six
// products per (out tile, chain tile, K-block) as in gauss_bf16x3.h, smallest first, two independent accumulators
// alternating; fp32 accumulation.  Two kernels:
//
//   gauss_res_langevin_kernel<OT>  (dims 164 .. 224; 132 .. 160 only with records -- the plain call there keeps Ps resident in LDS:
//     gauss_mfma.hip, five tiles)   the state STAYS IN REGISTERS (C/D layout, 16 OT registers + 16 OT
//     accumulators, one wave per SIMD, 128 chains per workgroup): no HBM traffic in the step loop; the B operand of a
//     K-block is eight state registers; the split work (B operands, next slab) sits in slots behind the MFMAs.
//   gauss_big_langevin_kernel<OT, NS>  (dims 228 .. 512)   a step is one pass of a tiled GEMM over the state, the
//     Euler-Maruyama update its epilogue: the state goes through HBM / L2 once per step (read as the B operand -- lane (m, h)
//     loads eight coordinates of chain m as two 16 B pieces --, read again by the epilogue, written once).  Up to 256
//     out-dims: eight waves of one chain tile each (two per SIMD, 256 registers: one's VALU work issues while the other's
//     MFMAs run), 256 chains per workgroup.  Beyond: two SLICES of <= 8 out tiles, four waves; the updated first slice
//     waits in registers until the second slice has read the old state (in place, no second state buffer).
//
// Where the time goes (MI355X, 2^17 chains x 256 dims x 20 steps; scripts/ab_big.sh removes one phase at a time,
// profiles/r03_ab_gauss_big.txt): the MFMAs 1.46 ms (0.82 ms of pipe time at the bf16 peak for the 6 products), Philox +
// Box-Muller + update 0.85 - 1.0 ms, the slab path 0.5 - 1.5 ms, not overlapped: one wave per SIMD overlaps only what is placed
// between two MFMAs by hand.  3.0 ms as shipped (round 2's lane-group kernel: 17.9 ms; the same chain as torch ops -- a GEMM and
// four element-wise kernels per step -- 8.3 ms).  Of the slab path the split and the LDS writes ARE hidden (slots); what is left is
// the ISSUE of its global loads -- ~250 cycles per global_load_dwordx4 and wave, independent of footprint (all loads aliased to
// 1 KB: same time), of the lead (half a stage or a whole one) and nearly of coalescing (-5 %).  Open: fewer, wider requests for Ps
// (LDS-direct buffer loads of the fp32 slab would move the split to the readers: 4x the split work), and the normals behind the
// MFMAs as in gauss_mfma.hip's FAST body (16 OT more registers).
#pragma once
#include "ebm_common.h"
#include "diag.h"
#include "gauss_bf16x3.h"
#include "mlp_b16.h"  // EBM_BLOCK_CUT

namespace ebm {
namespace gbig {

using gauss3::bf16x8;
using gauss3::f32x16;
using gauss3::f32x8;
using gauss3::static_for;

#ifndef EBM_BIG_WAVES1
#define EBM_BIG_WAVES1 8  /* waves per workgroup of the one-slice kernels: 8 = two per SIMD, one chain tile each; 4 = see EBM_BIG_TWO_WG */
#endif
#ifndef EBM_BIG_TWO_WG
#define EBM_BIG_TWO_WG 0  /* with 4 waves: 1 = one chain tile per wave, one K-block per stage, TWO workgroups per CU; 0 = two tiles per wave */
#endif
#ifndef EBM_BIG_EXP
#define EBM_BIG_EXP 0  /* timing experiments (scripts/ab_big.sh): 1 no MFMA, 2 no Philox, 4 no Ps loads, 8 no slab work, 16 no stage barrier, 32 no B splits, 64 slab split fed from state registers, 256 / 512 loads aliased to 8 rows / one column block */
#endif

struct BigArgs {
  float* x;
  int64_t n_chains;
  int32_t dim, k_steps;
  float eta, sqrt_eta, noise_coef;
  const float4* table;
  const float* noise;
  int32_t clamp_on;
  float cmin, cmax;
  int32_t thin;
  int64_t n_kept;
  float* traj;
  const float* mean;
  const float* prec;
  RngKey key;
  uint64_t step0;
  float* energy_out;    // k_steps == 0: evaluation only (ebm_energy_grad_f32) -- E [n] and / or the gradient [n, dim], either may be null;
  float* grad_out;      // x is read, not written
  const char* prec_image;  // IMG kernels: Ps pre-split in slab order (ebm_gauss_prec_image_f32; ebm_energy_t.aux), else null
  diag::DiagArgs diag;  // records of the kept steps (diag.h: one per wave-tile of 32 chains, E = 32 dim, S = dim); partials == nullptr: none
  int32_t sh_classes = 1;       // SHIFTED rows (the resident kernel's SH instantiations: widths off multiples of 4): alignment classes,
  int64_t sh_image_stride = 0;  // and the bytes between the classes' images in prec_image
};

struct Tri {
  bf16x8 h, m, l;
};
__device__ __forceinline__ Tri split8(const f32x8 d) {
  Tri t;
  t.h = __builtin_convertvector(d, bf16x8);
  const f32x8 r1 = d - __builtin_convertvector(t.h, f32x8);
  t.m = __builtin_convertvector(r1, bf16x8);
  const f32x8 r2 = r1 - __builtin_convertvector(t.m, f32x8);
  t.l = __builtin_convertvector(r2, bf16x8);
  return t;
}
// (native vectors throughout: a conditional on HIP's float4 STRUCT is compiled through a stack slot)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t gauss3_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x8 join8(const f32x4 a, const f32x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
template <int OT, int NS>
struct BigCfg {
  // One slice: EIGHT waves of one chain tile each -- two waves per SIMD, 256 registers each (128 accumulators), so that one wave's
  // VALU work (splits, Philox, update) issues while the other's MFMAs run; a single wave per SIMD issues VALU work at ~40 % of
  // the rate and nothing overlaps (scripts/ab_big.sh: the phases add up).  Two slices need 256 accumulator registers: four waves.
  static constexpr int WAVES = NS == 1 ? EBM_BIG_WAVES1 : 4;
  static constexpr int THREADS = 64 * WAVES;
  static constexpr bool TWO_WG = NS == 1 && WAVES == 4 && EBM_BIG_TWO_WG;
  static constexpr int CTW = NS == 1 && !TWO_WG ? 8 / WAVES : 1;  // chain tiles per wave
  static constexpr int KBS = TWO_WG ? 1 : 2;              // K-blocks (of 16 columns) per stage
  static constexpr int CHAINS = 32 * WAVES * CTW;         // per workgroup
  static constexpr int UNITS = OT * 64 * KBS;            // lane-operand units of a slab: [OT][KBS K-blocks][64 lanes]
  static constexpr int UPT = (UNITS + THREADS - 1) / THREADS;
  static constexpr size_t SLAB = (size_t)3 * UNITS * 16;  // bytes of one buffer (three splits)
  static constexpr size_t SMEM = 2 * SLAB + 512 * sizeof(float);
};

// bytes of the image of a [dim x dim] matrix for the (OT, NS) kernel: [NS][n_stage][3][UNITS][16]
template <int OT, int NS>
constexpr size_t big_image_bytes(int dim) {
  using C = BigCfg<OT, NS>;
  return (size_t)NS * (size_t)(((dim + 31) & ~31) / (16 * C::KBS)) * 3u * C::UNITS * 16u;
}
// IMG (round 4): the slabs come READY-MADE.  ebm_gauss_prec_image_f32 lays the three bf16 pieces of Ps out once, in global
// memory, in the order the stages consume them -- [slice][stage][piece][unit], each stage's 3 UNITS 16 bytes exactly the LDS
// image store_a() would have written -- and a stage's slab is moved by LDS-direct loads (global_load_lds_dwordx4: 1 KiB per
// wave-instruction, no registers, no split, no ds_write): six to twelve instructions per wave and stage where the fp32 path
// spends ~100 per thread on loads, masks, splits and stores.  The image is 1.5 x the fp32 matrix (0.4 - 1.5 MB: L2-resident).
template <int OT, int NS, bool DIAG = false, bool IMG = false>
__global__ __launch_bounds__((BigCfg<OT, NS>::THREADS), (BigCfg<OT, NS>::TWO_WG ? 2 : 1)) void gauss_big_langevin_kernel(BigArgs a) {
  using C = BigCfg<OT, NS>;
  constexpr int kBigBlock = C::THREADS;
  constexpr int CTW = C::CTW, UNITS = C::UNITS, UPT = C::UPT, KBS = C::KBS, KW = 16 * KBS;
  extern __shared__ __align__(16) unsigned char big_smem[];
  bf16x8* slab = reinterpret_cast<bf16x8*>(big_smem);                   // [2][3][UNITS]
  float* mus = reinterpret_cast<float*>(big_smem + 2 * C::SLAB);        // [d32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
  const int dim = a.dim, d32 = (dim + 31) & ~31, n_stage = d32 / KW;
  for (int i = tid; i < d32; i += kBigBlock) mus[i] = i < dim ? a.mean[i] : 0.0f;

  int64_t chain[CTW];
  bool active[CTW];
  int64_t xoff[CTW];  // element offset of the chain's row (row 0 for lanes past the last chain: never stored).  An OFFSET, not a
                      // pointer: the per-step launder below would strip a pointer of its address space (flat loads count in lgkmcnt too)
#pragma unroll
  for (int c = 0; c < CTW; ++c) {
    chain[c] = (int64_t)blockIdx.x * C::CHAINS + (wave * CTW + c) * 32 + m;
    active[c] = chain[c] < a.n_chains;
    xoff[c] = (active[c] ? chain[c] : 0) * (int64_t)dim;
  }
  __syncthreads();

  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;
  int until_keep = a.thin;
  int64_t kept = 0;

  // One lane-operand unit of the slab of slice rows `row0`, stage s: eight fp32 of a row of Ps.  Loads are UNCONDITIONAL from
  // an always-valid address and the padding is zeroed where the value is consumed, a stage later (mask_*): a select right
  // behind the load makes the compiler wait for every load before it issues the next one.
  auto a_ok = [&](int row0, int s, int j, bool& ok0, bool& ok1, int& row, int& kcol) {
    const int u = tid + kBigBlock * j;
    const int it = u / (64 * KBS), kb2 = (u >> 6) % KBS, ul = u & 63;
    row = row0 + 32 * it + (ul & 31);
    kcol = KW * s + 16 * kb2 + 8 * (ul >> 5);
    const bool ok = u < UNITS && row < dim;
    ok0 = ok && kcol < dim;
    ok1 = ok && kcol + 4 < dim;
  };
  auto load_a = [&](int row0, int s, int j, f32x4& v0, f32x4& v1) {
    bool ok0, ok1;
    int row, kcol;
    a_ok(row0, s, j, ok0, ok1, row, kcol);
    const float* p = a.prec + (int64_t)row * dim + kcol;
    v0 = *reinterpret_cast<const f32x4*>(ok0 ? p : a.prec);
    v1 = *reinterpret_cast<const f32x4*>(ok1 ? p + 4 : a.prec);
  };
  auto store_a = [&](int buf, int row0, int s, int j, const f32x4& v0, const f32x4& v1) {  // (row0, s): what was loaded
    const int u = tid + kBigBlock * j;
    if (u < UNITS) {
      bool ok0, ok1;
      int row, kcol;
      a_ok(row0, s, j, ok0, ok1, row, kcol);
      const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
      const Tri t = split8(join8(ok0 ? v0 : z, ok1 ? v1 : z));
      bf16x8* dst = slab + (size_t)buf * 3 * UNITS + u;
      dst[0] = t.h; dst[UNITS] = t.m; dst[2 * UNITS] = t.l;
    }
  };
  // IMG: request the slab of (slice sl, stage s) into buffer `buf` -- the pieces of 1 KiB dealt round-robin to the waves.
  // Assembly on purpose: behind the builtin the compiler puts s_waitcnt vmcnt(0) in front of every later LDS read (it
  // assumes they alias the transfer); the one wait that is needed stands in front of the stage barrier.
  [[maybe_unused]] const uint32_t slab_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)big_smem;
  [[maybe_unused]] auto dma_stage = [&](int buf, int sl, int s) {
    constexpr uint32_t STAGE_BYTES = 3u * UNITS * 16u;
    const char* src = a.prec_image + (size_t)(sl * n_stage + s) * STAGE_BYTES;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t dst = slab_lds + (uint32_t)buf * STAGE_BYTES;
    for (int piece = wv; piece < (int)(STAGE_BYTES / 1024u); piece += C::WAVES) {
      const uint32_t voff = (uint32_t)(piece * 1024 + lane * 16), base = dst + (uint32_t)piece * 1024u;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(src), "s"(base) : "memory");
    }
  };
  auto load_b = [&](int c, int s, int kb2, f32x4& v0, f32x4& v1) {
    const int kcol = KW * s + 16 * kb2 + 8 * h;
    const float* p = a.x + xoff[c] + kcol;
    v0 = *reinterpret_cast<const f32x4*>(kcol < dim ? p : a.x);
    v1 = *reinterpret_cast<const f32x4*>(kcol + 4 < dim ? p + 4 : a.x);
  };
  auto masked_b = [&](int c, int s, int kb2, const f32x4& v0, const f32x4& v1) {
    const int kcol = KW * s + 16 * kb2 + 8 * h;
    const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
    return join8((active[c] && kcol < dim) ? v0 : z, (active[c] && kcol + 4 < dim) ? v1 : z);
  };

  // Records (DIAG; as in the resident kernel below): column sums of a kept state from the epilogue's registers; its energy
  // 0.5 d . P d one step late, from the next step's contraction and the old state the epilogue reads anyway; a kept LAST step
  // costs one more trip (contraction and energy only: `upd` false).
  [[maybe_unused]] int rec_keep = 0, rec_pending = -1;
  const bool records = DIAG && a.diag.partials != nullptr;
  const bool eval_only = DIAG && a.k_steps == 0;  // one contraction: gradient and energy out, nothing updated
  if (eval_only) rec_pending = 0;
  const int n_trips = eval_only ? 1 : a.k_steps + ((records && a.k_steps > 0 && a.k_steps % a.thin == 0) ? 1 : 0);
  for (int step = 0; step < n_trips; ++step) {
    const bool upd = !DIAG || step < a.k_steps;
    if (a.table && upd) {
      const float4 tb = a.table[step];
      eta = tb.x; sqrt_eta = tb.y; noise_coef = tb.z;
    }
    const bool keep_now = a.traj && until_keep == 1 && upd;
    const bool rec_now = records && until_keep == 1 && upd;
    [[maybe_unused]] float e_acc[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) e_acc[c] = 0.0f;
    // (hidden from LICM: left visible, every quad's address of every step is formed before the step loop and spilled)
    int h4 = 4 * h;
    asm volatile("" : "+v"(h4));
    uint64_t e_rows[CTW];
#pragma unroll
    for (int c = 0; c < CTW; ++c) {
      asm volatile("" : "+v"(xoff[c]));
      e_rows[c] = (uint64_t)chain[c] * (uint64_t)dim;
      asm volatile("" : "+v"(e_rows[c]));
    }
    f32x16 res[NS][CTW][OT];  // accumulators of a slice, then its updated state

    static_for<NS>([&](auto slc) {
      constexpr int sl = decltype(slc)::value;
      constexpr int row0 = sl * 32 * OT;
      static_for<CTW * OT>([&](auto ic) {
        constexpr int c = decltype(ic)::value / OT, ot = decltype(ic)::value % OT;
#pragma unroll
        for (int r = 0; r < 16; ++r) res[sl][c][ot][r] = 0.0f;
      });

      // ---- stage 0 of the slice: its slab and this wave's B operands
      [[maybe_unused]] f32x4 ra[IMG ? 1 : UPT][2];
      f32x4 rb[CTW][KBS][2];
      if constexpr (IMG) dma_stage(0, sl, 0);
      else static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; load_a(row0, 0, j, ra[j][0], ra[j][1]); });
      static_for<CTW * KBS>([&](auto ic) {
        constexpr int c = decltype(ic)::value / KBS, kb2 = decltype(ic)::value % KBS;
        load_b(c, 0, kb2, rb[c][kb2][0], rb[c][kb2][1]);
      });
      if constexpr (IMG) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; store_a(0, row0, 0, j, ra[j][0], ra[j][1]); });
      __syncthreads();

      for (int s = 0; s < n_stage; ++s) {
        const int buf = s & 1;
        const bool more = s + 1 < n_stage;
        // this stage's B operands first (their loads had the previous stage to land), THEN the next stage's loads: the
        // compiler's wait for a loop-carried load is counted from the newest one in flight
        Tri ball[KBS][CTW];
        static_for<KBS * CTW>([&](auto ic) {
          constexpr int kb2 = decltype(ic)::value / CTW, c = decltype(ic)::value % CTW;
          const f32x4 m0 = *reinterpret_cast<const f32x4*>(mus + KW * s + 16 * kb2 + 8 * h);
          const f32x4 m1 = *reinterpret_cast<const f32x4*>(mus + KW * s + 16 * kb2 + 8 * h + 4);
          ball[kb2][c] = split8(masked_b(c, s, kb2, rb[c][kb2][0], rb[c][kb2][1]) - join8(m0, m1));
        });
        if (more && !(EBM_BIG_EXP & 4)) {  // a whole stage of matrix work for them to land
          if constexpr (IMG) dma_stage(buf ^ 1, sl, s + 1);
          else static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; load_a(row0, s + 1, j, ra[j][0], ra[j][1]); });
          static_for<CTW * KBS>([&](auto ic) {
            constexpr int c = decltype(ic)::value / KBS, kb2 = decltype(ic)::value % KBS;
            load_b(c, s + 1, kb2, rb[c][kb2][0], rb[c][kb2][1]);
          });
        }
        __builtin_amdgcn_sched_barrier(0);
        const bf16x8* sb = slab + (size_t)buf * 3 * UNITS + lane;
        static_for<KBS>([&](auto kc) {
          constexpr int kb2 = decltype(kc)::value;
          const Tri (&b)[CTW] = ball[kb2];
          // pairs of independent accumulators alternate: (two chain tiles, one A triple) or (one chain tile, two out tiles)
          constexpr int PAIRS = CTW == 2 ? OT : (OT + 1) / 2;
          // the A triples of pair p + 1 are requested before the twelve MFMAs of pair p (fenced: left alone, the scheduler
          // hoists every ds_read of the stage to its top -- 6 OT operand registers per K-block)
          auto read_a = [&](auto pc, bf16x8 (&a6)[6]) {
            constexpr int pi = decltype(pc)::value;
            constexpr int ot0 = CTW == 2 ? pi : 2 * pi, ot1 = CTW == 2 ? pi : (2 * pi + 1 < OT ? 2 * pi + 1 : 2 * pi);
            a6[0] = sb[2 * UNITS + ot0 * (64 * KBS) + kb2 * 64]; a6[1] = sb[UNITS + ot0 * (64 * KBS) + kb2 * 64]; a6[2] = sb[ot0 * (64 * KBS) + kb2 * 64];
            if constexpr (CTW == 1 && ot1 != ot0) {
              a6[3] = sb[2 * UNITS + ot1 * (64 * KBS) + kb2 * 64]; a6[4] = sb[UNITS + ot1 * (64 * KBS) + kb2 * 64]; a6[5] = sb[ot1 * (64 * KBS) + kb2 * 64];
            } else {
              a6[3] = a6[0]; a6[4] = a6[1]; a6[5] = a6[2];
            }
          };
          bf16x8 acur[6];
          read_a(std::integral_constant<int, 0>{}, acur);
          static_for<PAIRS>([&](auto pc) {
            constexpr int pi = decltype(pc)::value;
            constexpr int ot0 = CTW == 2 ? pi : 2 * pi, ot1 = CTW == 2 ? pi : (2 * pi + 1 < OT ? 2 * pi + 1 : 2 * pi);
            constexpr int c1 = CTW == 2 ? 1 : 0;
            constexpr bool two = CTW == 2 || 2 * pi + 1 < OT;
            bf16x8 anext[6];
            if constexpr (pi + 1 < PAIRS) read_a(std::integral_constant<int, pi + 1>{}, anext);
            __builtin_amdgcn_sched_barrier(0);
            f32x16 g0 = res[sl][0][ot0], g1;
            if constexpr (two) g1 = res[sl][c1][ot1];
            const Tri& b0 = b[0];
            const Tri& b1 = b[c1];
            if constexpr (!(EBM_BIG_EXP & 1)) {
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[0], b0.h, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[3], b1.h, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[1], b0.m, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[4], b1.m, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[1], b0.h, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[4], b1.h, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[2], b0.l, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[5], b1.l, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[2], b0.m, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[5], b1.m, g1, 0, 0, 0);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[2], b0.h, g0, 0, 0, 0);
            if constexpr (two) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[5], b1.h, g1, 0, 0, 0);
            } else {
              g0[0] += (float)acur[0][0] * (float)b0.h[0] + (float)acur[1][0] * (float)b0.m[0] + (float)acur[2][0] * (float)b0.l[0];
              if constexpr (two) g1[0] += (float)acur[3][0] * (float)b1.h[0] + (float)acur[4][0] * (float)b1.m[0] + (float)acur[5][0] * (float)b1.l[0];
            }
            res[sl][0][ot0] = g0;
            if constexpr (two) res[sl][c1][ot1] = g1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (pi + 1 < PAIRS) {
#pragma unroll
              for (int i = 0; i < 6; ++i) acur[i] = anext[i];
            }
          });
        });
        if constexpr (IMG) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next slab (and its next B operands) have landed
        } else if (more && !(EBM_BIG_EXP & 8)) {
          static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; store_a(buf ^ 1, row0, s + 1, j, ra[j][0], ra[j][1]); });
        }
        __syncthreads();  // the next slab is written, this one is read by everyone
      }

      // ---- Euler-Maruyama update of the slice in the reference's op order (one Philox counter per register quad).
      // The old state of tile t + 2 is requested while tile t is updated: a quad's load alone is an HBM round trip.
      constexpr int TILES = CTW * OT, AHEAD = 2;
      f32x4 xold[AHEAD + 1][4];
      auto request = [&](auto tc) {
        constexpr int t = decltype(tc)::value, c = t / OT, ot = t % OT;
        static_for<4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const int d0 = row0 + 32 * ot + 8 * q + h4;
          int off = (active[c] && d0 < dim) ? d0 : 0;
          asm volatile("" : "+v"(off));  // pins the load here (volatile asm keeps its order: the block cuts, the other quads)
          xold[t % (AHEAD + 1)][q] = *reinterpret_cast<const f32x4*>(a.x + xoff[c] + off);  // (not ok: some valid word, never stored)
        });
      };
      static_for<(AHEAD < TILES ? AHEAD : TILES)>([&](auto tc) { request(tc); });
      static_for<TILES>([&](auto ic) {
        constexpr int t = decltype(ic)::value, c = t / OT, ot = t % OT;
        const uint64_t e_row = e_rows[c];
        if constexpr (t + AHEAD < TILES) request(std::integral_constant<int, t + AHEAD>{});
        static_for<4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const int d0 = row0 + 32 * ot + 8 * q + h4;
          const bool ok = active[c] && d0 < dim;
          const int off = ok ? d0 : 0;
          const f32x4 xo = xold[t % (AHEAD + 1)][q];
          if constexpr (DIAG) {
            if (rec_pending >= 0) {  // the energy share of the state kept one step ago: (x - mu) . g, before g is overwritten
              const f32x4 mq = *reinterpret_cast<const f32x4*>(mus + (ok ? d0 : 0));
#pragma unroll
              for (int i = 0; i < 4; ++i) e_acc[c] = __builtin_fmaf(ok ? xo[i] - mq[i] : 0.0f, res[sl][c][ot][4 * q + i], e_acc[c]);
            }
            if (eval_only && ok && a.grad_out) {
              const f32x4 gv = {res[sl][c][ot][4 * q], res[sl][c][ot][4 * q + 1], res[sl][c][ot][4 * q + 2], res[sl][c][ot][4 * q + 3]};
              *reinterpret_cast<f32x4*>(a.grad_out + xoff[c] + d0) = gv;
            }
          }
          f32x4 eps = {0.0f, 0.0f, 0.0f, 0.0f};
          if (!upd) {
            // the extra trip of a kept last step: no draw, no update
          } else if constexpr (EBM_BIG_EXP & 2) {
            eps = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
          } else if (a.noise) {
            eps = *reinterpret_cast<const f32x4*>(a.noise + (int64_t)step * a.n_chains * dim + (active[c] ? (int64_t)e_row : 0) + off);
          } else {
            const F4 n4 = normal4_at(a.key, (e_row + (uint64_t)d0) >> 2, a.step0 + (uint64_t)step);
            eps = f32x4{n4.v[0], n4.v[1], n4.v[2], n4.v[3]};
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float x1 = xo[i] - eta * res[sl][c][ot][4 * q + i];
            const float dw = eps[i] * sqrt_eta;
            float nv = x1 + noise_coef * dw;
            if (a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
            res[sl][c][ot][4 * q + i] = nv;
          }
          if constexpr (sl == NS - 1) {  // nobody reads the old state after the last slice's K loop: store at once
            if (ok && upd) {
              const f32x4 v = {res[sl][c][ot][4 * q], res[sl][c][ot][4 * q + 1], res[sl][c][ot][4 * q + 2], res[sl][c][ot][4 * q + 3]};
              *reinterpret_cast<f32x4*>(a.x + xoff[c] + d0) = v;
              if (keep_now) *reinterpret_cast<f32x4*>(a.traj + ((int64_t)e_row * a.n_kept + kept * (int64_t)dim) + d0) = v;
            }
          }
          __builtin_amdgcn_sched_barrier(0);  // one Philox call's temporaries at a time
        });
        if constexpr (DIAG) {
          if (rec_now)  // this tile of the kept state (padding rows hold garbage only where `c < dim` fails: not stored)
            diag::wave_record<1>(a.diag.partials, a.diag.n_blocks, rec_keep, (int64_t)blockIdx.x * (C::CHAINS / 32) + wave * CTW + c, dim,
                                 [&](int, int r) { return res[sl][c][ot][r]; }, active[c], lane, sl * OT + ot);
        }
        EBM_BLOCK_CUT();  // one tile per basic block: the scheduler does not stretch 64 Philox calls over each other
      });
    });
    if constexpr (DIAG) {
      if (rec_pending >= 0) {
#pragma unroll
        for (int c = 0; c < CTW; ++c) {
          float acc = e_acc[c];
          acc += __shfl_xor(acc, 32);
          if (eval_only) {
            if (a.energy_out && active[c] && h == 0) a.energy_out[chain[c]] = 0.5f * acc;
          } else
          diag::wave_record_tail(a.diag.partials, a.diag.n_blocks, rec_pending, (int64_t)blockIdx.x * (C::CHAINS / 32) + wave * CTW + c, dim,
                                 0.5f * acc, active[c], false, lane);
        }
        rec_pending = -1;
      }
      if (!upd) break;
      if (rec_now) rec_pending = rec_keep++;
    }

    // ---- the held slices: every slice has read the old state by now
    if constexpr (NS > 1) {
      int h4s = 4 * h;
      asm volatile("" : "+v"(h4s));  // (fresh conditions: shared with the epilogue's, 2 SGPRs per quad stay live across a slice)
      static_for<(NS - 1) * CTW * OT>([&](auto ic) {
        constexpr int sl = decltype(ic)::value / (CTW * OT), c = (decltype(ic)::value / OT) % CTW, ot = decltype(ic)::value % OT;
        constexpr int row0 = sl * 32 * OT;
        static_for<4>([&](auto qc) {
          constexpr int q = decltype(qc)::value;
          const int d0 = row0 + 32 * ot + 8 * q + h4s;
          if (active[c] && d0 < dim) {
            const f32x4 v = {res[sl][c][ot][4 * q], res[sl][c][ot][4 * q + 1], res[sl][c][ot][4 * q + 2], res[sl][c][ot][4 * q + 3]};
            *reinterpret_cast<f32x4*>(a.x + xoff[c] + d0) = v;
            if (keep_now) *reinterpret_cast<f32x4*>(a.traj + ((int64_t)e_rows[c] * a.n_kept + kept * (int64_t)dim) + d0) = v;
          }
        });
      });
    }
    if (--until_keep == 0) {
      until_keep = a.thin;
      ++kept;
    }
    // (a wave reads only its own chains' rows, as B operands and in its epilogue: its stores above are complete before
    //  its loads of the next step -- no workgroup barrier is needed for x)
    __builtin_amdgcn_s_waitcnt(0);
  }
}

// ---------------------------------------------------------------------------------
// dims 132 .. 256: the chain state stays in registers for the whole call (as below 129: C/D layout, quad q of tile t =
// coordinates 32 t + 8 q + 4 h .. + 3 of chain m) and only Ps streams -- per stage of two K-blocks the workgroup stages the
// [32 OT] x 32 slab of Ps as three operand-ready bf16 images in LDS (double-buffered, one barrier per stage); the B operand
// of K-block kb is eight consecutive state registers (gauss_bf16x3.h: lane half h supplies coordinates
// 16 kb + 8 (j >> 2) + 4 h + (j & 3), and the slab unit of lane (row, h) holds Ps[row] at the same columns).  No HBM
// traffic in the step loop: the state is read once and written once per call (plus the kept rows of a trajectory);
// Ps crosses L2 -> CU once per workgroup (128 chains) and step.  One wave per SIMD: 16 OT state + 16 OT accumulator registers.
// ---------------------------------------------------------------------------------
// A three-way split of eight values in EIGHT steps of five / six instructions (mlp_b16.h: pair p = elements 2 p, 2 p + 1 = one
// packed dword of each piece): step 2 p forms the hi piece of pair p and its residual, step 2 p + 1 the mid and lo pieces.
template <bool PK>
struct SplitJobT {
  f32x8 d;
  mlpb16::f32x2 r;
  mlpb16::Split8p t;
  template <class K>
  __device__ __forceinline__ void step(K) {
    constexpr int k = K::value, pr = k >> 1;
    // (PK = false: element-wise arithmetic -- behind an MFMA a packed-f32 instruction stalls the wave ~20 cycles,
    //  profiles/r06_mfma_valu_overlap.txt: the resident Langevin kernel, -1.3 % at dim 256; the HMC transition kernels of
    //  gauss_stream_e.h keep the packed form -- they spill, and the longer element-wise stream cost them 3 - 12 %)
    if constexpr ((k & 1) == 0) {
      mlpb16::pair_split_a<pr, PK>(t, r, mlpb16::f32x2{d[2 * pr], d[2 * pr + 1]});
    } else {
      mlpb16::pair_split_b<pr, PK>(t, r);
      mlpb16::pair_split_c<pr>(t, r);
    }
  }
  __device__ __forceinline__ Tri tri() const {
    Tri o;
    o.h = __builtin_bit_cast(bf16x8, t.h); o.m = __builtin_bit_cast(bf16x8, t.m); o.l = __builtin_bit_cast(bf16x8, t.l);
    return o;
  }
};
typedef SplitJobT<true> SplitJob;

template <int OT>
struct ResCfg {
  static constexpr int THREADS = 256;
  static constexpr int UNITS = OT * 128;                  // [OT][2 K-blocks][64 lanes]
  static constexpr int UPT = OT;                          // 16 B chunks of the slab per thread and stage: one per pass of 32 rows
  static constexpr int SLABU = OT * 128;                  // units per image
  static constexpr size_t SLAB = (size_t)3 * SLABU * 16;
  static constexpr size_t SMEM = 2 * SLAB + 256 * sizeof(float);
};

// IMG (round 4): as in the tiled kernel, the three images of a stage arrive ready-made from the pre-split copy of Ps
// (ebm_gauss_prec_image_f32 appends this kernel's layout -- [stage][piece][unit], the rotated slots included -- behind the
// tiled kernel's, see res_image_offset) by LDS-direct loads: the 5 UPT chunk steps per thread and stage (mask, split, three
// 8-byte writes: ~50 instructions per chunk, the larger half of a stage's non-matrix work) and the UPT load registers go.
template <int OT>
constexpr size_t res_image_bytes() { return (size_t)OT * 3u * ResCfg<OT>::SLABU * 16u; }

// FOLD (round 4; re-done in round 6 -- with IMG, plain call: no records, in-kernel draws): the step's normals are drawn BEHIND THE
// MFMAs, in the slots the slab split left empty, and folded into the state in place -- x[t] += noise_coef (eps sqrt_eta) as soon
// as tile t has served as a B operand for the last time -- so the epilogue is x - eta g and no register holds a normal across the
// contraction's end.  The sum is associated (x + noise) - eta g instead of the reference's (x - eta g) + noise
// (core/base_integrator.py:711-731): the same three terms, each product rounded as there, one rounding in a different place --
// inside the tolerance these kernels are held to (bf16 x 3 contraction); EVERY instantiation of this kernel (records, injected
// noise, shifted rows) uses that association, so the native draws stay bit-identical to the materialised field and records do
// not change the samples.
// Round 4's version bought 1 - 5 % and was left off ("one wave per SIMD is bound by the instructions it issues, not by where they
// stand").  What the pinned-stream measurements of round 6 say (profiles/r06_mfma_valu_overlap.txt): FIVE plain vector
// instructions hide behind a 32-cycle MFMA and every further one costs its issue time -- and round 4's stages were 6 (a Philox
// round with its key bump) to ~20 instructions (a Box-Muller pair with its fold), the folds piled up behind the last tiles (a tile's
// draws waited for the tile to stop being an operand), and the MFMA shared a scheduling region with its slot.  Now: per quad one
// FOLD stage + 21 DRAW stages of <= 5 instructions (counter | ten Philox rounds | two Box-Muller pairs in five stages each, the
// scaled normals parked in four registers per quad); the draws run LAG = 4 quads (one tile) AHEAD of the folds, so only a fold waits
// for its tile; a slot takes ceil(stages left / free slots left) stages; the MFMA is fenced from its slot.
template <int OT>
struct FoldPlan {
  static constexpr int HALF = 6 * OT, SLOTS = 2 * HALF * OT, B_STEPS = 10;
  static constexpr int LAG = 4, NQ = 4 * OT, PER_QUAD = 22, NST = PER_QUAD * (NQ + LAG);
  int pos[SLOTS + 1];  // stages taken before slot i (slot = stage * 2 HALF + ordinal)
  static constexpr bool is_free(int s, int o) {
    if (o >= HALF) return true;
    if (o / 2 >= B_STEPS) return true;
    return (o % 2 == 1) && !(s + 1 < OT);
  }
  // stage K: block b = K / 22; r == 0: the fold of quad b - LAG; r = 1 .. 21: draw stage r - 1 of quad b
  static constexpr bool is_noop(int K) {
    const int b = K / PER_QUAD, r = K % PER_QUAD;
    return r == 0 ? b < LAG : b >= NQ;
  }
  static constexpr bool allowed(int K, int s, int o) {  // a fold waits until its tile is no longer a B operand
    const int b = K / PER_QUAD, r = K % PER_QUAD;
    if (r != 0 || b < LAG) return true;
    return (b - LAG) / 4 < (o >= HALF ? s + 1 : s);
  }
  constexpr FoldPlan() : pos{} {
    int free_left = 0, real_left = 0;
    for (int i = 0; i < SLOTS; ++i) free_left += is_free(i / (2 * HALF), i % (2 * HALF)) ? 1 : 0;
    for (int K = 0; K < NST; ++K) real_left += is_noop(K) ? 0 : 1;
    int p = 0;
    for (int i = 0; i < SLOTS; ++i) {
      pos[i] = p;
      const int s = i / (2 * HALF), o = i % (2 * HALF);
      if (!is_free(s, o)) continue;
      int want = (real_left + free_left - 1) / free_left;
      if (i + 1 == SLOTS) want = real_left;  // (the last slot takes whatever is left: the last tile's folds)
      while (p < NST && (is_noop(p) || (want > 0 && allowed(p, s, o)))) {
        if (!is_noop(p)) { --want; --real_left; }
        ++p;
      }
      --free_left;
    }
    pos[SLOTS] = p;
  }
};

// SH: SHIFTED rows (gauss_mfma_body.h says how) -- widths off multiples of 4 whose rows reach 161 .. 256 tile coordinates: a
// workgroup takes the chains of one alignment class and streams that class's image (of the shifted matrix).
template <int OT, bool DIAG = false, bool IMG = false, bool FOLD = false, bool SH = false>
__global__ __launch_bounds__(256) void gauss_res_langevin_kernel(BigArgs a) {
  static_assert(!FOLD || (IMG && !DIAG), "FOLD is the plain call on the image");
  static_assert(!SH || IMG, "shifted rows: on the per-class images");  // (round 6: FOLD too -- a folded draw on a padding coordinate is zeroed by the update's select)
  using C = ResCfg<OT>;
  constexpr int UPT = C::UPT, SLABU = C::SLABU;
  extern __shared__ __align__(16) unsigned char big_smem[];
  bf16x8* slab = reinterpret_cast<bf16x8*>(big_smem);                   // [2][3][UNITS]
  float* mus = reinterpret_cast<float*>(big_smem + 2 * C::SLAB);        // [32 OT]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, m = lane & 31, h = lane >> 5;
  const int dim = a.dim;
  const int sh_s = SH ? (int)(blockIdx.x % (unsigned)a.sh_classes) : 0;
  const int lo = SH ? ((dim * sh_s) & 3) : 0, hi = lo + dim;  // the row's tile coordinates
  [[maybe_unused]] const char* image = a.prec_image + (SH ? (int64_t)sh_s * a.sh_image_stride : 0);
  for (int i = tid; i < 32 * OT; i += 256) mus[i] = (i >= lo && i < hi) ? a.mean[i - lo] : 0.0f;
  const int64_t chain = SH ? (((int64_t)(blockIdx.x / (unsigned)a.sh_classes) * 4 + wave) * 32 + m) * a.sh_classes + sh_s
                           : ((int64_t)blockIdx.x * 4 + wave) * 32 + m;
  const bool active = chain < a.n_chains;
  const int64_t row = active ? chain * (int64_t)dim - lo : 0;  // flat element of tile coordinate 0

  f32x16 x[OT];
  static_for<OT * 4>([&](auto ic) {
    constexpr int t = decltype(ic)::value >> 2, q = decltype(ic)::value & 3;
    const int k0 = 32 * t + 8 * q + 4 * h;
    if constexpr (SH) {
      f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
      if (active && k0 >= lo && k0 + 3 < hi) {
        v = *reinterpret_cast<const f32x4*>(a.x + row + k0);
      } else if (active && k0 + 3 >= lo && k0 < hi) {  // a quad shared with a neighbouring chain: its own elements only
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (k0 + i >= lo && k0 + i < hi) v[i] = a.x[row + k0 + i];
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) x[t][4 * q + i] = v[i];
    } else {
    const bool ok = active && k0 < dim;
    const f32x4 v = *reinterpret_cast<const f32x4*>(a.x + row + (ok ? k0 : 0));
#pragma unroll
    for (int i = 0; i < 4; ++i) x[t][4 * q + i] = ok ? v[i] : 0.0f;  // padding stays exactly 0: zero rows / columns of Ps, never stored
    }
  });

  // The slab of a stage: rows 0 .. 32 OT - 1 of Ps at the stage's 32 columns -- one 128 B line per row, EIGHT lanes per line:
  // lane (tid & 7) = chunk c loads the four columns 32 s + 4 c .. + 3 of row 32 j + (tid >> 3) in pass j (one fully
  // coalesced instruction per pass: 8 rows x 128 B; as 16 B pieces strided by the row length every piece was its own request
  // to the L1 -- 2 048 per stage and CU -- and the slab took more than a stage to arrive whatever the lead: 40 % of the step).
  // A chunk is HALF a lane-operand unit: unit (row, kb2 = c >> 2, h' = c & 1) holds the columns 16 kb2 + 4 h' + {0..3} and
  // + 8 that lane half h' pairs with its B registers in K-block 2 s + kb2; chunk c is its half (c >> 1) & 1.  The split is
  // element-wise, so every lane splits its own four values and writes 8 B of the unit's slot in each of the three images.
  // The slots of a (tile, kb2, h') group are ROTATED by 2 (2 kb2 + h'): the eight lanes of a row then write to eight
  // different 8 B bank groups (unrotated, the four units of a row alias: 512 B apart); the b128 reads follow the rotation.
  const int ch = tid & 7, rrow = tid >> 3;                       // chunk, row inside a pass
  const int ch_kb2 = ch >> 2, ch_h = ch & 1, ch_half = (ch >> 1) & 1;
  const int wr_unit = ch_kb2 * 64 + ch_h * 32 + ((rrow + 2 * (2 * ch_kb2 + ch_h)) & 31);
  auto chunk_ok = [&](int s, int j) { return 32 * j + rrow < dim && 32 * s + 4 * ch < dim; };
  auto load_a = [&](int s, int j, f32x4& v) {
    const int rr = (EBM_BIG_EXP & 256) ? ((32 * j + rrow) & 7) : (32 * j + rrow);   // experiment 256: every pass reads the same 8 rows
    const int ss = (EBM_BIG_EXP & 512) ? 0 : s;                                       // experiment 512: every stage reads the same columns
    const float* p = a.prec + (int64_t)rr * dim + 32 * ss + 4 * ch;
    v = *reinterpret_cast<const f32x4*>(chunk_ok(s, j) ? p : a.prec);
  };
  typedef uint32_t u32x2_t __attribute__((ext_vector_type(2)));
  auto write_a = [&](int buf, int j, const mlpb16::Split8p& t) {  // pairs 0, 1 of t: this chunk's four values
    unsigned char* dst = reinterpret_cast<unsigned char*>(slab + (size_t)buf * 3 * SLABU + j * 128 + wr_unit) + 8 * ch_half;
    *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{t.h[0], t.h[1]};
    *reinterpret_cast<u32x2_t*>(dst + (size_t)SLABU * 16) = u32x2_t{t.m[0], t.m[1]};
    *reinterpret_cast<u32x2_t*>(dst + (size_t)2 * SLABU * 16) = u32x2_t{t.l[0], t.l[1]};
  };
  // one chunk in five steps: mask | pair 0 hi | pair 0 mid, lo | pair 1 hi | pair 1 mid, lo + the three 8 B writes
  struct ChunkJob {
    f32x4 d;
    mlpb16::f32x2 r;
    mlpb16::Split8p t;
  };
  auto chunk_step = [&](ChunkJob& cj, int buf, int s_of, auto jc, auto kc, const f32x4& v) {
    constexpr int j = decltype(jc)::value, k = decltype(kc)::value;
    if constexpr (k == 0) {
      const f32x4 z = {0.0f, 0.0f, 0.0f, 0.0f};
      cj.d = chunk_ok(s_of, j) ? v : z;
    } else if constexpr (k == 1) {
      mlpb16::pair_split_a<0>(cj.t, cj.r, mlpb16::f32x2{cj.d[0], cj.d[1]});
    } else if constexpr (k == 2) {
      mlpb16::pair_split_b<0>(cj.t, cj.r);
      mlpb16::pair_split_c<0>(cj.t, cj.r);
    } else if constexpr (k == 3) {
      mlpb16::pair_split_a<1>(cj.t, cj.r, mlpb16::f32x2{cj.d[2], cj.d[3]});
    } else {
      mlpb16::pair_split_b<1>(cj.t, cj.r);
      mlpb16::pair_split_c<1>(cj.t, cj.r);
      write_a(buf, j, cj.t);
    }
  };
  const int rd_unit[2] = {h * 32 + ((m + 2 * h) & 31), 64 + h * 32 + ((m + 2 * (2 + h)) & 31)};  // this lane's operand slot per K-block

  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;
  int until_keep = a.thin;
  int64_t kept = 0;
  Tri b0;  // the B operand of the next K-block 0
  int gstage = 0;  // stages done so far: its parity is the LDS buffer (OT may be odd, the pipeline runs across steps)

  // IMG: request stage s of the image into buffer `buf` (6 OT pieces of 1 KiB dealt round-robin to the four waves; assembly so
  // that the compiler does not serialise every later LDS read behind the transfer -- see the tiled kernel)
  [[maybe_unused]] const uint32_t slab_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)big_smem;
  [[maybe_unused]] auto dma_stage = [&](int buf, int s) {
    constexpr uint32_t STAGE_BYTES = 3u * SLABU * 16u;
    const char* src = image + big_image_bytes<OT, 1>(32 * OT) + (size_t)s * STAGE_BYTES;
    const int wv = __builtin_amdgcn_readfirstlane(wave);
    const uint32_t dst = slab_lds + (uint32_t)buf * STAGE_BYTES;
    for (int piece = wv; piece < (int)(STAGE_BYTES / 1024u); piece += 4) {
      const uint32_t voff = (uint32_t)(piece * 1024 + lane * 16), base = dst + (uint32_t)piece * 1024u;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(src), "s"(base) : "memory");
    }
  };
  // ... one request of it (this wave's i-th: piece wave + 4 i), for the slots: the CU's texture-address unit takes a 1 KiB request
  // every ~16 cycles and a wave that issues its 9 .. 12 in a row stands in that queue with the matrix pipe idle (round 6,
  // scripts/hmc_stream_phase_times.py on the HMC kernels that share this stage loop) -- one behind every fourth MFMA of K-block 0
  [[maybe_unused]] auto dma_piece = [&](int buf, int s, auto ic) {
    constexpr uint32_t STAGE_BYTES = 3u * SLABU * 16u;
    constexpr int i = decltype(ic)::value;
    const char* src = image + big_image_bytes<OT, 1>(32 * OT) + (size_t)s * STAGE_BYTES;
    const int piece = __builtin_amdgcn_readfirstlane(wave) + 4 * i;
    if (4 * (i + 1) <= (int)(STAGE_BYTES / 1024u) || piece < (int)(STAGE_BYTES / 1024u)) {
      const uint32_t voff = (uint32_t)(piece * 1024 + lane * 16), base = slab_lds + (uint32_t)buf * STAGE_BYTES + (uint32_t)piece * 1024u;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(src), "s"(base) : "memory");
    }
  };
  // the first slab
  [[maybe_unused]] f32x4 ra[IMG ? 1 : UPT];
  if constexpr (IMG) {
    dma_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
  static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; load_a(0, j, ra[j]); });
  static_for<UPT>([&](auto jc) {
    ChunkJob cj;
    static_for<5>([&](auto kc) { chunk_step(cj, 0, 0, jc, kc, ra[decltype(jc)::value]); });
  });
  // (from here on `ra` holds the slab the NEXT stage splits: a chunk's registers are reloaded -- for the slab after that --
  //  as soon as its split has copied them, a full stage before they are needed again)
  static_for<UPT>([&](auto jc) { constexpr int j = decltype(jc)::value; load_a(1 % OT, j, ra[j]); });
  }
  __syncthreads();

  // Records (DIAG): the column sums of a kept state come from the registers right after its update; its energy 0.5 d . P d
  // is what the NEXT step's contraction computes (g = P d), so the energy share of a record is written one step late and only
  // a kept LAST step costs a contraction of its own (one more trip of the loop, without an update).
  // (SH: the records of the classes interleave -- record (group, class), diag.h plan_classes)
  const int64_t wave_id = SH ? ((int64_t)(blockIdx.x / (unsigned)a.sh_classes) * 4 + wave) * a.sh_classes + sh_s : (int64_t)blockIdx.x * 4 + wave;
  [[maybe_unused]] int rec_keep = 0, rec_pending = -1;
  const int n_trips = a.k_steps + ((DIAG && a.k_steps > 0 && a.k_steps % a.thin == 0) ? 1 : 0);
  for (int step = 0; step < n_trips; ++step) {
    if (a.table && step < a.k_steps) {
      const float4 tb = a.table[step];
      eta = tb.x; sqrt_eta = tb.y; noise_coef = tb.z;
    }
    f32x16 g[OT];
    static_for<OT>([&](auto tc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) g[decltype(tc)::value][r] = 0.0f;
    });
    int tid_s = tid;
    asm volatile("" : "+v"(tid_s));  // (per step: the slab addresses are not hoisted out of the step loop and spilled)
    // B operand of K-block (tile t, half kb2): d = x[t][8 kb2 ..] - mu, then its three-way split in five small steps -- the
    // steps are SLOTS behind the MFMAs (one wave per SIMD: work placed between two MFMAs issues while the first one runs;
    // placed before or after the MFMA block it adds its full issue time, ~40 % of the step as measured by scripts/ab_big.sh)
    typedef SplitJobT<false> ResSplit;  // (element-wise: its steps sit behind MFMAs)
    auto b_init = [&](ResSplit& jb, auto tc, auto kc, auto hc) {  // half hc of the eight differences
      constexpr int t = decltype(tc)::value, kb2 = decltype(kc)::value, hf = decltype(hc)::value;
      const f32x4 mm = *reinterpret_cast<const f32x4*>(mus + 32 * t + 16 * kb2 + 8 * hf + 4 * h);
#pragma unroll
      for (int j = 0; j < 4; ++j) jb.d[4 * hf + j] = x[t][8 * kb2 + 4 * hf + j] - mm[j];
    };
    {  // K-block 0 of the step: its state registers were written by the previous update
      ResSplit j0;
      b_init(j0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      b_init(j0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      static_for<8>([&](auto kc) { j0.step(kc); });
      b0 = j0.tri();
    }
    // FOLD: the noise stages of this step (see FoldPlan): one quad's Philox state in flight, the scaled normals of LAG quads parked
    [[maybe_unused]] uint32_t nc0 = 0, nc1 = 0, nc2 = 0, nc3 = 0, nk0 = 0, nk1 = 0;
    [[maybe_unused]] float nz[FoldPlan<OT>::LAG][4];
    [[maybe_unused]] float bu = 0.0f, brev = 0.0f, br = 0.0f, bs = 0.0f;
    [[maybe_unused]] uint64_t n_row = (uint64_t)chain * (uint64_t)dim - (uint64_t)lo;  // (SH: tile coordinate 0 of the chain's shifted row)
    if constexpr (FOLD) asm volatile("" : "+v"(n_row));
    [[maybe_unused]] auto noise_stage = [&](auto kc) {
      using P = FoldPlan<OT>;
      constexpr int K = decltype(kc)::value, blk = K / P::PER_QUAD, r = K % P::PER_QUAD;
      if constexpr (r == 0) {  // the fold of quad blk - LAG: its tile stopped being a B operand (FoldPlan::allowed)
        if constexpr (blk >= P::LAG) {
          constexpr int n = blk - P::LAG, t = n / 4, q = n % 4;
          float f0 = x[t][4 * q] + nz[n % P::LAG][0], f1 = x[t][4 * q + 1] + nz[n % P::LAG][1];
          float f2 = x[t][4 * q + 2] + nz[n % P::LAG][2], f3 = x[t][4 * q + 3] + nz[n % P::LAG][3];
          // (every stage ends in a volatile asm on what it wrote: the stages are pure arithmetic whose results are wanted much later,
          //  and without an ordering point the instruction selector's linearisation collects them at the end of the block -- round
          //  4's fold sat in 8 clumps behind the stages' last MFMAs, scripts/isa_gapmap.py -- instead of behind the MFMAs they were
          //  written behind)
          asm volatile("" : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3));
          x[t][4 * q] = f0; x[t][4 * q + 1] = f1; x[t][4 * q + 2] = f2; x[t][4 * q + 3] = f3;
        }
      } else if constexpr (blk < P::NQ) {
        constexpr int t = blk / 4, q = blk % 4, sub = r - 1;  // draw stage sub = 0 .. 20 of quad (t, q)
        if constexpr (sub == 0) {
          const uint64_t grp = (n_row + (uint64_t)(32 * t + 8 * q + 4 * h)) >> 2;
          const uint64_t stp = a.step0 + (uint64_t)step;
          nc0 = (uint32_t)grp; nc1 = (uint32_t)(grp >> 32); nc2 = (uint32_t)stp; nc3 = (uint32_t)(stp >> 32);
          nk0 = a.key.k0; nk1 = a.key.k1;
          asm volatile("" : "+v"(nc0), "+v"(nc1));
        } else if constexpr (sub <= 10) {  // one round of philox4x32_10 (ebm_common.h); the key schedule is uniform (scalar unit)
          const uint64_t p0 = (uint64_t)0xD2511F53u * nc0;
          const uint64_t p1 = (uint64_t)0xCD9E8D57u * nc2;
          const uint32_t n0 = xor3((uint32_t)(p1 >> 32), nc1, nk0);
          const uint32_t n2 = xor3((uint32_t)(p0 >> 32), nc3, nk1);
          nc1 = (uint32_t)p1; nc3 = (uint32_t)p0; nc0 = n0; nc2 = n2;
          nk0 += 0x9E3779B9u; nk1 += 0xBB67AE85u;
          asm volatile("" : "+v"(nc0), "+v"(nc1), "+v"(nc2), "+v"(nc3));
        } else {  // box_muller (ebm_common.h) on (nc0, nc1), then on (nc2, nc3), in five stages each; the same operations
          constexpr int pr = (sub - 11) / 5, st = (sub - 11) % 5;
          if constexpr (st == 0) {
            bu = u01_open_low(pr == 0 ? nc0 : nc2);
            brev = (float)(pr == 0 ? nc1 : nc3) * 0x1p-32f;
            asm volatile("" : "+v"(bu), "+v"(brev));
          } else if constexpr (st == 1) {
            br = -1.38629436111989061883f * __builtin_amdgcn_logf(bu);
            asm volatile("" : "+v"(br));
          } else if constexpr (st == 2) {
            br = __builtin_amdgcn_sqrtf(br);
            bs = __builtin_amdgcn_sinf(brev);
            asm volatile("" : "+v"(br), "+v"(bs));
          } else if constexpr (st == 3) {
            brev = br * __builtin_amdgcn_cosf(brev);  // n1
            bs = br * bs;                              // n0
            asm volatile("" : "+v"(brev), "+v"(bs));
          } else {
            float z0 = noise_coef * (bs * sqrt_eta), z1 = noise_coef * (brev * sqrt_eta);
            asm volatile("" : "+v"(z0), "+v"(z1));
            nz[blk % P::LAG][2 * pr] = z0;
            nz[blk % P::LAG][2 * pr + 1] = z1;
          }
        }
      }
    };
    // the A operands of unit (kb2, pi) from the slab at `base`: [piece][out tile][128 units]; read_a_part: piece 2 - i (low first, as
    // the MFMAs use them) of both tiles
    auto read_a = [&](const bf16x8* base, auto kc, auto pc, bf16x8 (&a6)[6]) {
      constexpr int kb2 = decltype(kc)::value, pi = decltype(pc)::value, ot0 = 2 * pi, ot1 = 2 * pi + 1 < OT ? 2 * pi + 1 : 2 * pi;
      const bf16x8* sr = base + rd_unit[kb2];
      a6[0] = sr[2 * SLABU + ot0 * 128]; a6[1] = sr[SLABU + ot0 * 128]; a6[2] = sr[ot0 * 128];
      if constexpr (ot1 != ot0) {
        a6[3] = sr[2 * SLABU + ot1 * 128]; a6[4] = sr[SLABU + ot1 * 128]; a6[5] = sr[ot1 * 128];
      }
    };
    auto read_a_part = [&](const bf16x8* base, auto kc, auto pc, auto ic, bf16x8 (&a6)[6]) {
      constexpr int kb2 = decltype(kc)::value, pi = decltype(pc)::value, i = decltype(ic)::value;
      constexpr int ot0 = 2 * pi, ot1 = 2 * pi + 1 < OT ? 2 * pi + 1 : 2 * pi;
      const bf16x8* sr = base + rd_unit[kb2];
      a6[i] = sr[(2 - i) * SLABU + ot0 * 128];
      if constexpr (ot1 != ot0) a6[3 + i] = sr[(2 - i) * SLABU + ot1 * 128];
    };
    bf16x8 acur[6];
    if constexpr (IMG) read_a(slab + (size_t)(gstage & 1) * 3 * SLABU, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, acur);
    static_for<OT>([&](auto sc) {
      constexpr int s = decltype(sc)::value, sn = (s + 1) % OT;  // the stage after the last one is stage 0 of the next step
      constexpr int HALF = 6 * OT;                                // MFMAs per K-block
      const int buf = gstage & 1;
      constexpr int sn2 = (s + 2) % OT;  // the slab requested during this stage
#ifdef EBM_RES_DMA_BLOCK
      if constexpr (IMG) dma_stage(buf ^ 1, sn);  // (the barrier that ended the stage before freed that buffer)
#endif
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8* sb = slab + (size_t)buf * 3 * SLABU;
      ResSplit jb1, jb0n;
      // Slots: one behind every MFMA, each <= 6 .. 8 instructions (what fits a 32-cycle MFMA; more delays the next one).
      //   behind K-block 0 (HALF slots): the B operands of K-block 1 and of the next stage's K-block 0, alternating --
      //   2 init + 8 split steps each;   behind K-block 1: the next slab (its loads were issued at the top of the stage) --
      //   per unit 1 mask step, 8 split steps, the LDS write with the last one
      constexpr int A_STEPS = 5 * UPT, B_STEPS = 10;
      constexpr int A_PER = (A_STEPS + HALF - 1) / HALF;
      static_assert(2 * B_STEPS <= HALF && A_PER == 1, "the split work of a stage fits behind its MFMAs");
      auto b_job = [&](ResSplit& jb, auto tc, auto kc, auto kk) {  // step kk of 10 of a B operand
        constexpr int k = decltype(kk)::value;
        if constexpr (k < 2) b_init(jb, tc, kc, std::integral_constant<int, k>{});
        else jb.step(std::integral_constant<int, k - 2>{});
      };
      ChunkJob cj;
      auto a_job = [&](auto kk) {  // step kk of 5 UPT of the next slab
        constexpr int k = decltype(kk)::value, j = k / 5, st = k % 5;
        if constexpr ((EBM_BIG_EXP & 64) != 0 && st == 0) {  // timing experiment: the split's input from state registers
          cj.d = f32x4{x[s][0], x[s][1], x[s][2], x[s][3]};
        } else {
          chunk_step(cj, buf ^ 1, sn, std::integral_constant<int, j>{}, std::integral_constant<int, st>{}, ra[j]);
        }
        if constexpr (st == 0 && !(EBM_BIG_EXP & 4)) load_a(sn2, j, ra[j]);  // the copy above freed them
      };
      auto slot = [&](auto oc) {
        constexpr int o = decltype(oc)::value;
        if constexpr (FOLD) {
          constexpr FoldPlan<OT> plan{};
          constexpr int i = s * 2 * HALF + o, k0 = plan.pos[i], k1 = plan.pos[i + 1];
          static_for<k1 - k0>([&](auto kk) { noise_stage(std::integral_constant<int, k0 + decltype(kk)::value>{}); });
        }
#ifndef EBM_RES_DMA_BLOCK
        // the next slab's requests (the barrier that ended the stage before freed its buffer): 1.5 OT per wave, HALF / 4 gaps
        if constexpr (IMG && o < HALF && o % 4 == 1) dma_piece(buf ^ 1, sn, std::integral_constant<int, o / 4>{});
#endif
        if constexpr (o < HALF) {
          if constexpr (EBM_BIG_EXP & 32) {
          } else if constexpr (o % 2 == 0 && o / 2 < B_STEPS) {
            b_job(jb1, sc, std::integral_constant<int, 1>{}, std::integral_constant<int, o / 2>{});
          } else if constexpr (o % 2 == 1 && o / 2 < B_STEPS && s + 1 < OT) {
            b_job(jb0n, std::integral_constant<int, (s + 1 < OT ? s + 1 : 0)>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, o / 2>{});
          }
        } else if constexpr (!IMG) {
          constexpr int ak = (o - HALF) * A_PER;
          static_for<A_PER>([&](auto ic) {
            if constexpr (ak + decltype(ic)::value < A_STEPS && !(EBM_BIG_EXP & 8)) a_job(std::integral_constant<int, ak + decltype(ic)::value>{});
          });
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      // A unit = (K-block, pair of output tiles) = 12 / 6 MFMAs.  (round 6) The operands of unit u + 1 are requested behind the first
      // three MFMAs of unit u, TWO reads at a time -- an LDS read holds the wave's issue for 16 cycles: two behind an MFMA cost 8,
      // six in a row 96 with the matrix pipe idle (profiles/r06_mfma_valu_overlap.txt, ds128) -- and across the K-block boundary
      // too; only the stage's first unit reads in front of its MFMAs (its slab became visible at the barrier just passed).
      constexpr int PAIRS = (OT + 1) / 2, UNITS2 = 2 * PAIRS;
      // IMG (the slabs arrive by LDS-direct loads): the stage's ONE synchronisation point stands in front of its LAST unit, whose
      // operands are in registers by then -- wait for this wave's share of the next slab and for its own LDS reads, barrier: the
      // next slab is visible and nobody reads this one any more -- and the next stage's first operands are requested behind the
      // last unit's MFMAs (was: barrier at the stage's end, then six reads in front of the next stage's first MFMA).  A step's
      // first stage reads its first operands in front of its MFMAs (above the stage loop).
      const bf16x8* sbn = slab + (size_t)(buf ^ 1) * 3 * SLABU;
      if constexpr (!IMG) read_a(sb, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, acur);
      Tri bb = b0;
      static_for<UNITS2>([&](auto uc) {
        constexpr int u = decltype(uc)::value, kb2 = u / PAIRS, pi = u % PAIRS;
        constexpr int ot0 = 2 * pi, ot1 = 2 * pi + 1 < OT ? 2 * pi + 1 : 2 * pi;
        constexpr bool two = ot1 != ot0;
        constexpr int o0 = kb2 * HALF + 12 * pi;  // ordinal of this unit's first MFMA
        if constexpr (kb2 == 1 && pi == 0) bb = jb1.tri();
        if constexpr (IMG && u + 1 == UNITS2) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          if constexpr (!(EBM_BIG_EXP & 16)) __syncthreads();
        }
        bf16x8 anext[6];
        __builtin_amdgcn_sched_barrier(0);
        f32x16 g0 = g[ot0], g1;
        if constexpr (two) g1 = g[ot1];
        // (term, operand) in issue order: smallest products first
        static_for<6>([&](auto tc) {
          constexpr int term = decltype(tc)::value;
          constexpr int ai = term == 0 ? 0 : (term <= 2 ? 1 : 2);                    // Pl | Pm Pm | Ph Ph Ph
          const bf16x8& bp = (term == 0 || term == 2 || term == 5) ? bb.h : ((term == 1 || term == 4) ? bb.m : bb.l);
          if constexpr (!(EBM_BIG_EXP & 1)) g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[ai], bp, g0, 0, 0, 0);
          else g0[term] += (float)acur[ai][0] * (float)bp[0];
          __builtin_amdgcn_sched_barrier(0);  // (round 6: the MFMA first -- in one region with its slot it can sink below it)
          if constexpr (term < 3 && u + 1 < UNITS2)
            read_a_part(sb, std::integral_constant<int, (u + 1) / PAIRS>{}, std::integral_constant<int, (u + 1) % PAIRS>{}, tc, anext);
          else if constexpr (term < 3 && IMG && s + 1 < OT)
            read_a_part(sbn, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, tc, anext);
          slot(std::integral_constant<int, o0 + (two ? 2 : 1) * term>{});
          if constexpr (two) {
            if constexpr (!(EBM_BIG_EXP & 1)) g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[3 + ai], bp, g1, 0, 0, 0);
            else g1[term] += (float)acur[3 + ai][0] * (float)bp[0];
            __builtin_amdgcn_sched_barrier(0);
            slot(std::integral_constant<int, o0 + 2 * term + 1>{});
          }
        });
        g[ot0] = g0;
        if constexpr (two) g[ot1] = g1;
        if constexpr (u + 1 < UNITS2 || (IMG && s + 1 < OT)) {
#pragma unroll
          for (int i = 0; i < 6; ++i) acur[i] = anext[i];
        }
      });
      if constexpr (s + 1 < OT) b0 = jb0n.tri();
      ++gstage;
      if constexpr (!IMG && !(EBM_BIG_EXP & 16)) __syncthreads();  // the next slab is written, this one is read by everyone
      // (a block cut here -- the OT unrolled stages are ONE basic block -- was tried: more spills in the plain kernels, dim 224 2.31 -> 2.84 ms)
    });

    if constexpr (DIAG) {
      if (rec_pending >= 0) {  // the energy of the state kept one step ago
        float acc = 0.0f;
        static_for<OT * 4>([&](auto ic) {
          constexpr int t = decltype(ic)::value >> 2, q = decltype(ic)::value & 3;
          const f32x4 mq = *reinterpret_cast<const f32x4*>(mus + 32 * t + 8 * q + 4 * h);
#pragma unroll
          for (int i = 0; i < 4; ++i) acc = __builtin_fmaf(x[t][4 * q + i] - mq[i], g[t][4 * q + i], acc);
        });
        acc += __shfl_xor(acc, 32);
        diag::wave_record_tail(a.diag.partials, a.diag.n_blocks, rec_pending, wave_id, dim, 0.5f * acc, active, false, lane);
        rec_pending = -1;
      }
      if (step >= a.k_steps) break;  // the extra trip of a kept last step
    }
    // ---- Euler-Maruyama update in the reference's op order, one Philox counter per register quad; all in registers
    uint64_t e_row = (uint64_t)chain * (uint64_t)dim - (uint64_t)lo;
    asm volatile("" : "+v"(e_row));
    const bool keep_now = a.traj && until_keep == 1;
    static_for<OT>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      static_for<4>([&](auto qc) {
        constexpr int q = decltype(qc)::value;
        const int k0 = 32 * t + 8 * q + 4 * h;
        const bool ok = active && k0 < dim;
        f32x4 eps;
        if constexpr (FOLD) {
          eps = f32x4{0.0f, 0.0f, 0.0f, 0.0f};  // (already in x)
        } else if constexpr (EBM_BIG_EXP & 2) {
          eps = f32x4{0.1f, 0.2f, 0.3f, 0.4f};
        } else if (a.noise) {
          if constexpr (SH) {  // (an injected field restarts at step * n * dim: no alignment to count on)
            eps = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const float* nz = a.noise + (int64_t)step * a.n_chains * dim + (active ? (int64_t)e_row : 0) + k0;
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (active && k0 + i >= lo && k0 + i < hi) eps[i] = nz[i];
          } else
          eps = *reinterpret_cast<const f32x4*>(a.noise + (int64_t)step * a.n_chains * dim + (active ? (int64_t)e_row : 0) + (ok ? k0 : 0));
        } else {
          const F4 n4 = normal4_at(a.key, (e_row + (uint64_t)k0) >> 2, a.step0 + (uint64_t)step);
          eps = f32x4{n4.v[0], n4.v[1], n4.v[2], n4.v[3]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          // six tiles and more (dims 161 .. 256, where the plain call is the FOLD form): (x + noise) - eta g in EVERY instantiation
          // (see FoldPlan; the FOLD form has the noise in x already).  Five tiles (dims 132 .. 160: this kernel only serves the calls
          // with records there, the plain call is gauss_mfma.hip's LDS-resident kernel): the reference's (x - eta g) + noise, as there.
          static_assert(!FOLD || OT >= 6, "the folded sum is the convention of the widths whose plain call folds");
          float nv;
          if constexpr (OT >= 6) {
            float xn = x[t][4 * q + i];
            if constexpr (!FOLD) {
              const float dw = eps[i] * sqrt_eta;
              xn = xn + noise_coef * dw;
            }
            nv = xn - eta * g[t][4 * q + i];
          } else {
            const float x1 = x[t][4 * q + i] - eta * g[t][4 * q + i];
            const float dw = eps[i] * sqrt_eta;
            nv = x1 + noise_coef * dw;
          }
          if (a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
          if constexpr (SH) x[t][4 * q + i] = (active && k0 + i >= lo && k0 + i < hi) ? nv : 0.0f;
          else x[t][4 * q + i] = ok ? nv : 0.0f;  // padding held at 0
        }
        if constexpr (SH) {
          if (keep_now && active) {  // (a kept row starts wherever (chain * n_kept + kept) * dim falls)
            float* tr = a.traj + ((int64_t)chain * a.n_kept + kept) * (int64_t)dim + (k0 - lo);
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (k0 + i >= lo && k0 + i < hi) tr[i] = x[t][4 * q + i];
          }
        } else
        if (keep_now && ok) {
          const f32x4 v = {x[t][4 * q], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]};
          *reinterpret_cast<f32x4*>(a.traj + ((int64_t)e_row * a.n_kept + kept * (int64_t)dim) + k0) = v;
        }
        __builtin_amdgcn_sched_barrier(0);  // one Philox call's temporaries at a time
      });
      EBM_BLOCK_CUT();
    });
    if (--until_keep == 0) {
      until_keep = a.thin;
      ++kept;
      if constexpr (DIAG) {
        diag::wave_record<OT>(a.diag.partials, a.diag.n_blocks, rec_keep, wave_id, dim, [&](int t, int r) { return x[t][r]; }, active, lane, 0, lo);
        rec_pending = rec_keep++;
      }
    }
  }
  static_for<OT * 4>([&](auto ic) {
    constexpr int t = decltype(ic)::value >> 2, q = decltype(ic)::value & 3;
    const int k0 = 32 * t + 8 * q + 4 * h;
    if constexpr (SH) {
      if (active && k0 >= lo && k0 + 3 < hi) {
        const f32x4 v = {x[t][4 * q], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]};
        *reinterpret_cast<f32x4*>(a.x + row + k0) = v;
      } else if (active && k0 + 3 >= lo && k0 < hi) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (k0 + i >= lo && k0 + i < hi) a.x[row + k0 + i] = x[t][4 * q + i];
      }
    } else
    if (active && k0 < dim) {
      const f32x4 v = {x[t][4 * q], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]};
      *reinterpret_cast<f32x4*>(a.x + row + k0) = v;
    }
  });
}

// The plain call on the image draws its normals behind the MFMAs (FoldPlan; round 6: on).  -DEBM_BIG_NOFOLD: the A/B build.
#ifdef EBM_BIG_NOFOLD  // A/B builds: the plain call keeps its normals in the epilogue
constexpr bool kResFold = false;
#else
constexpr bool kResFold = true;
#endif
// the SH instantiations' launcher (gauss_res_shift.hip)
template <int OT>
int launch_res_shift(const BigArgs& a, hipStream_t st) {
  using C = ResCfg<OT>;
  static DeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_res_langevin_kernel<OT, false, true, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_res_langevin_kernel<OT, true, true, false, true>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
  }
  const int64_t blocks = ceil_div64(ceil_div64(a.n_chains, a.sh_classes), 128) * a.sh_classes;  // class-major inside blockIdx: b % K
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  if (a.diag.partials) hipLaunchKernelGGL((gauss_res_langevin_kernel<OT, true, true, false, true>), dim3((unsigned)blocks), dim3(256), C::SMEM, st, a);
  else if (kResFold && !a.noise) {  // the plain call: its normals drawn behind the MFMAs (FoldPlan), as on the aligned widths
    static DeviceOnce fold_once;
    if (fold_once.first())
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_res_langevin_kernel<OT, false, true, true, true>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
    hipLaunchKernelGGL((gauss_res_langevin_kernel<OT, false, true, true, true>), dim3((unsigned)blocks), dim3(256), C::SMEM, st, a);
  } else hipLaunchKernelGGL((gauss_res_langevin_kernel<OT, false, true, false, true>), dim3((unsigned)blocks), dim3(256), C::SMEM, st, a);
  return check_launch("ebm_langevin_chain_f32");
}

template <int OT, bool IMG>
int launch_res_as(const BigArgs& a, hipStream_t st) {
  using C = ResCfg<OT>;
  static DeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_res_langevin_kernel<OT, false, IMG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)C::SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_res_langevin_kernel<OT, true, IMG>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)C::SMEM);
  }
  const int64_t blocks = ceil_div64(a.n_chains, 128);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  if (a.diag.partials) hipLaunchKernelGGL((gauss_res_langevin_kernel<OT, true, IMG>), dim3((unsigned)blocks), dim3(256), C::SMEM, st, a);
  else hipLaunchKernelGGL((gauss_res_langevin_kernel<OT, false, IMG>), dim3((unsigned)blocks), dim3(256), C::SMEM, st, a);
  return check_launch("ebm_langevin_chain_f32");
}
template <int OT>
int launch_res_plain_img(const BigArgs& a, hipStream_t st) {  // the plain call on the image (no records instantiation beside it)
  using C = ResCfg<OT>;
  static DeviceOnce attr_once;
  if (attr_once.first())
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_res_langevin_kernel<OT, false, true, kResFold>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)C::SMEM);
  if (kResFold && a.noise) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: the EBM_BIG_FOLD build draws its own normals at this width");
  const int64_t blocks = ceil_div64(a.n_chains, 128);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  hipLaunchKernelGGL((gauss_res_langevin_kernel<OT, false, true, kResFold>), dim3((unsigned)blocks), dim3(256), C::SMEM, st, a);
  return check_launch("ebm_langevin_chain_f32");
}
template <int OT>
int launch_res(const BigArgs& a, hipStream_t st) {
  if (a.prec_image && (reinterpret_cast<uintptr_t>(a.prec_image) & 15) == 0) {
    if constexpr (OT >= 6) {
      if (kResFold && !a.diag.partials && !a.noise) return launch_res_plain_img<OT>(a, st);
    }
    return launch_res_as<OT, true>(a, st);
  }
  return launch_res_as<OT, false>(a, st);
}

// the IMG instantiations live in gauss_big_img.hip (their own translation unit: compiled in parallel)
template <int OT, int NS>
int launch_big_img(const BigArgs& a, hipStream_t st);
#define EBM_BIG_IMG_DECL(OTV, NSV) template <> int launch_big_img<OTV, NSV>(const BigArgs& a, hipStream_t st);
EBM_BIG_IMG_DECL(5, 1) EBM_BIG_IMG_DECL(6, 1) EBM_BIG_IMG_DECL(7, 1) EBM_BIG_IMG_DECL(8, 1)
EBM_BIG_IMG_DECL(5, 2) EBM_BIG_IMG_DECL(6, 2) EBM_BIG_IMG_DECL(7, 2) EBM_BIG_IMG_DECL(8, 2)
#undef EBM_BIG_IMG_DECL

template <int OT, int NS, bool IMG = false>
int launch_big(const BigArgs& a, hipStream_t st) {
  using C = BigCfg<OT, NS>;
  if constexpr (!IMG) {
    if (a.prec_image && (reinterpret_cast<uintptr_t>(a.prec_image) & 15) == 0) return launch_big_img<OT, NS>(a, st);
  }
  static DeviceOnce attr_once;
  if (attr_once.first()) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_big_langevin_kernel<OT, NS, false, IMG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_big_langevin_kernel<OT, NS, true, IMG>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::SMEM);
  }
  const int64_t blocks = ceil_div64(a.n_chains, C::CHAINS);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  // Two slices of six or more tiles: the records instantiation ALSO for the plain call -- its uniform branches (`upd`, the
  // record tests) cut the epilogue's basic blocks and it allocates 414 .. 512 registers without a spill where the plain
  // instantiation spills 59 .. 242: dims 384 / 512 4.16 / 6.95 -> 4.07 / 6.07 ms (same box; dim 320, five tiles: 3.01 -> 3.16, kept plain)
  constexpr bool kRecordsKernelAlways = NS == 2 && OT >= 6;
  if (a.diag.partials || kRecordsKernelAlways || a.k_steps == 0) hipLaunchKernelGGL((gauss_big_langevin_kernel<OT, NS, true, IMG>), dim3((unsigned)blocks), dim3(C::THREADS), C::SMEM, st, a);
  else hipLaunchKernelGGL((gauss_big_langevin_kernel<OT, NS, false, IMG>), dim3((unsigned)blocks), dim3(C::THREADS), C::SMEM, st, a);
  return check_launch("ebm_langevin_chain_f32");
}

}  // namespace gbig
}  // namespace ebm
