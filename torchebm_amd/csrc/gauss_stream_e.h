// (shared by gauss_hmc_stream.hip and matrix_hmc_diag.hip)
// HMC transitions for the dense Gaussian energy at dims 164 .. 256 (multiples of 4): the matrix-layout transition body
// (mfma_hmc_body.h: position, momentum and force in the C/D layout of 32 x 32 tiles, one wave per 32 chains) with an
// evaluation that STREAMS the precision matrix -- the three bf16 images of a stage (32 columns of Ps for all 32 NT rows)
// arrive ready-made from the pre-split copy of Ps (ebm_gauss_prec_image_f32: the resident Langevin kernel's layout,
// gauss_big_body.h) by LDS-direct loads, double-buffered, one stage ahead of the MFMAs that read them; the B operands are split
// from the position registers in slots behind the MFMAs.  Up to 160 dims the images stay resident in LDS (gauss_hmc_mfma.hip);
// beyond 256 the position and momentum of 32 chains no longer fit a wave's registers (the sampler's GEMM route).
// Round 6: the force comes in PIECES of output tiles (eval_tiles<T0, TN>: a pass over the stages each, requesting only the slab
// part the piece reads) which the body kicks into the momentum at once -- position, momentum and a whole force array are never
// live together (round 5: 1 045 spilled values per lane at eight tiles, 41 GB of scratch traffic per launch) -- and the stage loop is
// software-pipelined (one sync per stage in front of its last unit, operands a unit ahead): docs/design/gaussian_big.md "Round 6".
// Before round 4 these widths ran per transition as library GEMMs + element-wise kernels: dim 256, 2^17 chains, 5 transitions
// of 10 leapfrog steps 18.6 ms.
// Reference: samplers/hmc.py:201-315 (transition), integrators/leapfrog.py:116-187, core/base_model.py:181-210 (energy).
#pragma once
#include "mfma_hmc_body.h"
#include "gauss_big_body.h"

namespace ebm {
namespace {

// -DEBM_PHASE_TIMES (scripts/hmc_stream_phase_times.py only; build gauss_hmc_stream.hip alone with it): wave 0 of workgroup 0 adds
// up shader-clock differences per phase class in registers (no memory traffic in the loop; a stamp waits for the wave's LDS / scalar
// queue and fences the scheduler) and writes the sums once, after its 20th pass.
// -DEBM_ABL_NODMA / NOSYNC / NOSPLIT / NOMFMA / NOBAD: timing ablations (wrong results): no slab requests / no waits and barriers /
// a quarter of the operand split / no MFMAs / never hand over to the literal body.
#ifdef EBM_PHASE_TIMES
__device__ unsigned long long ebm_hmc_phase_log[16];
// class c: 0 entry (first operand) | 1 units in front of the sync | 2 sync | 3 last unit | 4 energy part | 5 between passes
#define EBM_HSTAMP(c)                                                                \
  do {                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                               \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime();                    \
    if (ph_last_) ph_acc_[c] += now_ - ph_last_;                                     \
    ph_last_ = now_;                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                               \
  } while (0)
#else
#define EBM_HSTAMP(c) do {} while (0)
#endif

template <int NT>
struct GaussStreamE {
  using C = gbig::ResCfg<NT>;
  static constexpr int SLABU = C::SLABU;
  static constexpr uint32_t STAGE_BYTES = 3u * SLABU * 16u;
#ifndef EBM_STREAM_BUFS
#define EBM_STREAM_BUFS 2
#endif
  // Two slab buffers.  -DEBM_STREAM_BUFS=3 (round 6, measured, off): slab s + 2 requested while stage s runs, into the buffer stage
  // s - 1 read (free since that stage's barrier; 3 x 48 KB at eight tiles -- with the means and the masses 154 of the CU's 160 KB),
  // its requests dealt over a whole stage's gaps instead of standing in the last unit behind the barrier that freed their target,
  // the sync waiting by count (vmcnt(n): vector memory operations complete in order).  Same times within the run-to-run spread
  // (3.05 / 3.33 / 4.24 / 5.41 against 3.04 / 3.35 / 4.00 / 5.56 ms at dims 164 / 192 / 224 / 256): the requests are not what the last
  // unit waits for.
  static constexpr int kBufs = EBM_STREAM_BUFS;
  static constexpr int kSlabFloats = (int)((uint32_t)kBufs * STAGE_BYTES / 4u);
  static constexpr int kLdsFloats = kSlabFloats + 32 * NT;  // [kBufs buffers][3 pieces][SLABU units] | mu
  static constexpr bool kEvalGivesEnergy = true;
  static constexpr bool kCarry = false;      // (no LDS left for a parked force: L + 1 evaluations per transition, as the reference)
  static constexpr bool kBlockVote = true;   // barriers inside eval()
#ifdef EBM_PHASE_TIMES
  unsigned long long ph_last_ = 0, ph_acc_[6] = {0, 0, 0, 0, 0, 0};
#endif
  int gstage = 0;  // stages done so far
  int gbuf = 0;    // the buffer the next stage reads (the pipeline runs across calls)

  // request stage s of the image into buffer `buf`: 6 NT pieces of 1 KiB dealt round-robin to the four waves (assembly: see
  // gauss_big_body.h -- behind the builtin the compiler serialises every later LDS read)
  // dma<T0, TN>: the part of the stage that output tiles T0 .. T0 + TN - 1 read -- per bf16 piece TN * 2 KiB in a row
  template <int T0 = 0, int TN = NT>
  __device__ static __forceinline__ void dma(const GaussHmcArgs& a, const float* lds, int buf, int s) {
    // (readfirstlane: out of line -- gauss_hmc_fallback -- the compiler no longer knows these are wave-uniform; free in a kernel)
    const uint64_t src_v = (uint64_t)(uintptr_t)(a.prec_image + gbig::big_image_bytes<NT, 1>(32 * NT) + (size_t)s * STAGE_BYTES);
    const char* src = (const char*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(src_v >> 32)) << 32) |
                                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)src_v));
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane(
        (int)((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)lds + (uint32_t)buf * STAGE_BYTES));
    for (int c = wv; c < 6 * TN; c += kBlock / 64) {
      const uint32_t at = (uint32_t)(c / (2 * TN)) * (uint32_t)(SLABU * 16) + (uint32_t)(T0 * 2048 + (c % (2 * TN)) * 1024);
      const uint32_t voff = at + (uint32_t)(lane * 16), base = dst + at;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(src), "s"(base) : "memory");
    }
  }
  // ... the same request, this wave's chunks I0 .. I0 + NI - 1 only (its i-th chunk: c = wave + 4 i): inside eval_tiles the requests
  // go out one or two behind an MFMA -- the CU's one texture-address unit takes a 1 KiB request every ~16 cycles, and a wave that
  // issues its six in a row stands in that queue with the matrix pipe idle (scripts/hmc_stream_phase_times.py: the unit that
  // carried the requests took 1 130 cycles for 384 of MFMAs)
  template <int T0, int TN, int I0, int NI>
  __device__ static __forceinline__ void dma_chunks(const GaussHmcArgs& a, const float* lds, int buf, int s) {
    const uint64_t src_v = (uint64_t)(uintptr_t)(a.prec_image + gbig::big_image_bytes<NT, 1>(32 * NT) + (size_t)s * STAGE_BYTES);
    const char* src = (const char*)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(src_v >> 32)) << 32) |
                                               (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)src_v));
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t dst = (uint32_t)__builtin_amdgcn_readfirstlane(
        (int)((uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)lds + (uint32_t)buf * STAGE_BYTES));
#pragma unroll
    for (int i = I0; i < I0 + NI; ++i) {
      const int c = wv + (kBlock / 64) * i;
      if ((kBlock / 64) * (i + 1) <= 6 * TN || c < 6 * TN) {  // (the first: known when compiled -- no branch)
        const uint32_t at = (uint32_t)(c / (2 * TN)) * (uint32_t)(SLABU * 16) + (uint32_t)(T0 * 2048 + (c % (2 * TN)) * 1024);
        const uint32_t voff = at + (uint32_t)(lane * 16), base = dst + at;
        uint32_t keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(voff), "s"(src), "s"(base) : "memory");
      }
    }
  }
  __device__ static __forceinline__ void stage(const GaussHmcArgs& a, float* lds, int lo) {  // shifted rows (a.prec_image: this class's)
    for (int i = threadIdx.x; i < 32 * NT; i += kBlock) lds[kSlabFloats + i] = (i >= lo && i - lo < a.dim) ? a.mean[i - lo] : 0.0f;
    dma(a, lds, 0, 0);
    dma(a, lds, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __device__ static __forceinline__ void stage(const GaussHmcArgs& a, float* lds) {
    for (int i = threadIdx.x; i < 32 * NT; i += kBlock) lds[kSlabFloats + i] = i < a.dim ? a.mean[i] : 0.0f;
    dma(a, lds, 0, 0);  // the first evaluation's first two stages (the body's barrier follows)
    dma(a, lds, 1, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __device__ static __forceinline__ float energy(const GaussHmcArgs&, const float*, const Tile<NT>&, int, int) { return 0.0f; }

  // g^T = Ps (x - mu)^T, E = 0.5 (x - mu) . g.  Stage s = the 32 columns 32 s .. of Ps = the two K-blocks whose B operands are
  // registers 0 .. 7 and 8 .. 15 of position tile s; six products per (out tile, K-block) as in gauss_bf16x3.h, smallest first.
  // eval_tiles<T0, TN>: the OUTPUT tiles T0 .. T0 + TN - 1 only (a full pass over the stages; the B operands are split again) --
  // gout[i] = g tile T0 + i, the return value that part of the energy (want_e = false: 0 -- the force only).  (round 6) The transition body takes the force in PIECES
  // (kPieces passes) and kicks the momentum piece by piece, so that position, momentum and a whole force array are never live together
  // (mfma_hmc_body.h, PW).
#ifndef EBM_PW_PIECES8
#define EBM_PW_PIECES8 3
#endif
#ifndef EBM_PW_PIECES7
#define EBM_PW_PIECES7 2
#endif
  static constexpr int kPieces = NT == 8 ? EBM_PW_PIECES8 : (NT == 7 ? EBM_PW_PIECES7 : 2);
  // piece pi = tiles piece_t0(pi) .. + piece_tn(pi) - 1: NT / kPieces tiles each, the first NT % kPieces pieces one more
  static constexpr int piece_tn(int pi) { return NT / kPieces + (pi < NT % kPieces ? 1 : 0); }
  static constexpr int piece_t0(int pi) { return pi * (NT / kPieces) + (pi < NT % kPieces ? pi : NT % kPieces); }
  static constexpr int piece_of(int t0) {  // the piece that starts at tile t0
    for (int pi = 0; pi < kPieces; ++pi)
      if (piece_t0(pi) == t0) return pi;
    return 0;
  }
  __device__ __forceinline__ float eval(const GaussHmcArgs& a, const float* lds, const Tile<NT>& x, Tile<NT>& g, int m, int h) {
    return eval_tiles<0, NT>(a, lds, x, g.t, m, h);
  }
  template <int T0, int TN>
  __device__ __forceinline__ float eval_tiles(const GaussHmcArgs& a, const float* lds, const Tile<NT>& x, f32x16 (&gout)[TN], int m, int h,
                                              bool want_e = true) {
    using gbig::Tri;
    using gauss3::bf16x8;
    using gauss3::static_for;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const bf16x8* slab = reinterpret_cast<const bf16x8*>(lds);
    const float* mus = lds + kSlabFloats;
    int hs = h, ms = m;
    asm volatile("" : "+v"(hs), "+v"(ms));  // (per call: nothing derived from the lane is hoisted out of the trajectory loop and spilled)
    const int rd_unit[2] = {hs * 32 + ((ms + 2 * hs) & 31), 64 + hs * 32 + ((ms + 2 * (2 + hs)) & 31)};  // this lane's operand slot per K-block
    static_for<TN>([&](auto tc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) gout[decltype(tc)::value][r] = 0.0f;
    });
    // A B operand -- the eight differences x - mu of K-block (t, kb2), split three ways -- in MS = 20 micro-steps of two to four
    // instructions (round 6: five plain vector instructions hide behind a 32-cycle MFMA, profiles/r06_mfma_valu_overlap.txt; the
    // steps of gbig::SplitJob were five to eight, packed, all behind the first K-block's MFMAs): steps 0 .. 3 the differences of
    // pair p (position registers come from the accumulation file: a read each), then per pair hi piece + its widened copy |
    // residual | mid piece + copy | residual and lo piece.  Element-wise arithmetic (a packed-f32 instruction in a gap costs ~20 cycles).
    struct MicroSplit {
      f32x4 mm[2];  // the means of the eight elements (read at the start of the half the job runs in, ahead of the A operands)
      gauss3::f32x8 d;
      mlpb16::f32x2 r, wide;
      mlpb16::Split8p t;
      __device__ __forceinline__ Tri tri() const {
        Tri o;
        o.h = __builtin_bit_cast(bf16x8, t.h); o.m = __builtin_bit_cast(bf16x8, t.m); o.l = __builtin_bit_cast(bf16x8, t.l);
        return o;
      }
    };
    constexpr int MS = 20;
    auto b_means = [&](MicroSplit& jb, auto tc, auto kc) {
      constexpr int t = decltype(tc)::value, kb2 = decltype(kc)::value;
      jb.mm[0] = *reinterpret_cast<const f32x4*>(mus + 32 * t + 16 * kb2 + 4 * hs);
      jb.mm[1] = *reinterpret_cast<const f32x4*>(mus + 32 * t + 16 * kb2 + 8 + 4 * hs);
    };
    // (pin: a volatile empty asm on what the step ends in -- the operand of the NEXT stage is wanted only in the next stage, and
    //  the instruction selector otherwise collects its whole split behind the stage's last MFMA: scripts/isa_gapmap.py)
    auto b_micro = [&](MicroSplit& jb, auto tc, auto kc, auto kk, auto pinc) {
      constexpr int t = decltype(tc)::value, kb2 = decltype(kc)::value, k = decltype(kk)::value;
      constexpr bool pin = decltype(pinc)::value;
      if constexpr (k < 4) {
        jb.d[2 * k] = x.t[t][8 * kb2 + 2 * k] - jb.mm[k >> 1][2 * (k & 1)];
        jb.d[2 * k + 1] = x.t[t][8 * kb2 + 2 * k + 1] - jb.mm[k >> 1][2 * (k & 1) + 1];
        if constexpr (pin) asm volatile("" : "+v"(jb.d[2 * k]), "+v"(jb.d[2 * k + 1]));
      } else {
        constexpr int pr = (k - 4) >> 2, ph = (k - 4) & 3;
        if constexpr (ph == 0) mlpb16::pair_split_piece<pr, 0>(jb.t, jb.wide, mlpb16::f32x2{jb.d[2 * pr], jb.d[2 * pr + 1]});
        else if constexpr (ph == 1) mlpb16::pair_split_resid<false>(jb.r, mlpb16::f32x2{jb.d[2 * pr], jb.d[2 * pr + 1]}, jb.wide);
        else if constexpr (ph == 2) mlpb16::pair_split_piece<pr, 1>(jb.t, jb.wide, jb.r);
        else {
          mlpb16::pair_split_resid<false>(jb.r, jb.r, jb.wide);
          mlpb16::pair_split_c<pr>(jb.t, jb.r);
        }
        if constexpr (pin) {
          if constexpr (ph == 0 || ph == 2) asm volatile("" : "+v"(jb.wide.x), "+v"(jb.wide.y));
          else if constexpr (ph == 1) asm volatile("" : "+v"(jb.r.x), "+v"(jb.r.y));
          else asm volatile("" : "+v"(jb.t.l[pr]));
        }
      }
    };
    constexpr int HALF = 6 * TN, PAIRS = (TN + 1) / 2, UNITS = 2 * PAIRS;  // a unit: (K-block, pair of output tiles) = 12 / 6 MFMAs
    // the A operands of unit (kb2, pi) from the slab at sb: [piece][out tile][128 units]
    auto read_a = [&](const bf16x8* sb, auto kc, auto pc, bf16x8 (&a6)[6]) {
      constexpr int kb2 = decltype(kc)::value, pi = decltype(pc)::value, ot0 = T0 + 2 * pi, ot1 = 2 * pi + 1 < TN ? T0 + 2 * pi + 1 : T0 + 2 * pi;
      const bf16x8* sr = sb + rd_unit[kb2];
      a6[0] = sr[2 * SLABU + ot0 * 128]; a6[1] = sr[SLABU + ot0 * 128]; a6[2] = sr[ot0 * 128];
      if constexpr (ot1 != ot0) {
        a6[3] = sr[2 * SLABU + ot1 * 128]; a6[4] = sr[SLABU + ot1 * 128]; a6[5] = sr[ot1 * 128];
      }
    };
    // ... and TWO of them at a time (part i of three): an LDS read holds the wave's issue for 16 cycles -- two behind an MFMA cost
    // 8, six in a row 96 with the matrix pipe idle for most of it (profiles/r06_mfma_valu_overlap.txt, ds128)
    auto read_a_part = [&](const bf16x8* sb, auto kc, auto pc, auto ic, bf16x8 (&a6)[6]) {
      constexpr int kb2 = decltype(kc)::value, pi = decltype(pc)::value, ot0 = T0 + 2 * pi, ot1 = 2 * pi + 1 < TN ? T0 + 2 * pi + 1 : T0 + 2 * pi;
      constexpr int i = decltype(ic)::value;  // piece 2 - i (low first, as the MFMAs use them) of both tiles
      const bf16x8* sr = sb + rd_unit[kb2];
      a6[i] = sr[(2 - i) * SLABU + ot0 * 128];
      if constexpr (ot1 != ot0) a6[3 + i] = sr[(2 - i) * SLABU + ot1 * 128];
    };
    EBM_HSTAMP(5);
    Tri b0;
    bf16x8 acur[6];
    {
      MicroSplit j0;
      b_means(j0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      read_a(slab + (size_t)gbuf * 3 * SLABU, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, acur);
      static_for<MS>([&](auto kk) { b_micro(j0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, kk, std::false_type{}); });
      b0 = j0.tri();
    }
    // The stage loop, software-pipelined (round 6).  On entry to stage s: its slab has landed and is visible to the workgroup, the
    // load of slab s + 1 is in flight, the A operands of the stage's first unit are in registers (requested behind the previous
    // stage's last unit; for a call's first stage just above).  The stage's ONE synchronisation point stands in front of its LAST
    // unit, whose operands were requested a unit earlier: wait for this wave's share of slab s + 1 and for its own LDS reads, barrier
    // -- now slab s + 1 is visible and nobody reads slab s any more -- then request slab s + 2 into the buffer just freed and the
    // next stage's first operands behind the last unit's MFMAs.  (Before: barrier at the stage's end, the first twelve operand reads
    // of every stage and K-block issued in front of the MFMA that needs them -- the matrix pipe idle for an LDS round trip 2 NT
    // times per pass.)  Slabs s + 1, s + 2 beyond this pass are the next piece's first two (T0N, TNN).
    static_for<NT>([&](auto sc) {
      constexpr int s = decltype(sc)::value;
      constexpr int PIN = (piece_of(T0) + 1) % kPieces;  // the next piece (the literal body's one piece -- TN = NT -- follows itself)
      constexpr int T0N = TN == NT ? 0 : piece_t0(PIN), TNN = TN == NT ? NT : piece_tn(PIN);
      const int buf = gbuf, bufn = buf + 1 == kBufs ? 0 : buf + 1;
      const int buf2 = kBufs == 3 ? (bufn + 1 == kBufs ? 0 : bufn + 1) : buf;  // where slab s + 2 goes
      const bf16x8* sb = slab + (size_t)buf * 3 * SLABU;
      const bf16x8* sbn = slab + (size_t)bufn * 3 * SLABU;
      constexpr int TNR = s + 2 < NT ? TN : TNN;                                // slab s + 2: its tiles,
      constexpr int NCH = (6 * TNR + kBlock / 64 - 1) / (kBlock / 64);         // ... the requests of a wave (the last one not every wave's)
      constexpr int NCH_ALL = 6 * TNR / (kBlock / 64);                          // ... those every wave makes
      constexpr int O_LAST = HALF + 12 * (PAIRS - 1);                           // ordinal of the last unit's first MFMA
      EBM_HSTAMP((s == 0 ? 0 : 3));
      MicroSplit jb1, jb0n;
      // behind MFMA o of the stage: the first K-block's gaps hold the B operand of the second K-block, the second K-block's gaps the
      // next stage's first -- dealt evenly over gaps 2 .. HALF - 2 of the half (the means are requested at the half's start; the last
      // gap's results would be the next MFMA's operands)
      auto slot = [&](auto oc) {
        constexpr int o = decltype(oc)::value, ol = o % HALF, G0 = 2, NG = HALF - 1 - G0, ix = ol - G0;
        constexpr int lo = (ix >= 0 && ix < NG) ? MS * ix / NG : 0, hi = (ix >= 0 && ix < NG) ? MS * (ix + 1) / NG : 0;
#ifdef EBM_ABL_NOSPLIT
        constexpr int hi2 = hi < 5 ? hi : (lo < 5 ? 5 : lo);
#else
        constexpr int hi2 = hi;
#endif
        static_for<(hi2 > lo ? hi2 - lo : 0)>([&](auto uc) {
          constexpr int k = lo + decltype(uc)::value;
          if constexpr (o < HALF) b_micro(jb1, sc, std::integral_constant<int, 1>{}, std::integral_constant<int, k>{}, std::false_type{});
          else if constexpr (s + 1 < NT)
            b_micro(jb0n, std::integral_constant<int, (s + 1 < NT ? s + 1 : 0)>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, k>{}, std::true_type{});
        });
        __builtin_amdgcn_sched_barrier(0);
      };
      Tri bb = b0;
      static_for<UNITS>([&](auto uc) {
        constexpr int u = decltype(uc)::value, kb2 = u / PAIRS, pi = u % PAIRS;
        constexpr int l0 = 2 * pi, l1 = 2 * pi + 1 < TN ? 2 * pi + 1 : 2 * pi;  // local (piece) tile indices
        constexpr bool two = l1 != l0;
        constexpr int o0 = kb2 * HALF + 12 * pi;  // ordinal of this unit's first MFMA
        if constexpr (pi == 0) {
          if constexpr (kb2 == 0) b_means(jb1, sc, std::integral_constant<int, 1>{});
          else {
            if constexpr (s + 1 < NT) b_means(jb0n, std::integral_constant<int, (s + 1 < NT ? s + 1 : 0)>{}, std::integral_constant<int, 0>{});
            bb = jb1.tri();
          }
        }
        if constexpr (u + 1 == UNITS) {
          EBM_HSTAMP(1);
#ifndef EBM_ABL_NOSYNC
          // three buffers: this stage's requests (slab s + 2) are in flight behind slab s + 1's -- vector memory operations complete
          // in order, so "at most as many outstanding as every wave has requested since" says slab s + 1 has landed
          if constexpr (kBufs == 3) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" :: "n"(NCH_ALL) : "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __syncthreads();
#endif
          EBM_HSTAMP(2);
        }
        // the request for slab s + 2, this wave's chunks.  Three buffers: dealt over the gaps in FRONT of the sync (all of them
        // issued before it: the count it waits for); two: into the buffer the barrier just freed, over the last unit's gaps
        auto request = [&](auto jc) {
#ifndef EBM_ABL_NODMA
          constexpr int j = decltype(jc)::value;  // ordinal within the unit
          if constexpr (kBufs == 3) {
            constexpr int o = o0 + j;
            if constexpr (u + 1 < UNITS) {
              constexpr int i0 = (o * NCH + O_LAST - 1) / O_LAST, i1 = ((o + 1) * NCH + O_LAST - 1) / O_LAST;  // chunks i with i O_LAST / NCH in [o, o + 1)
              if constexpr (i1 > i0) {
                if constexpr (s + 2 < NT) dma_chunks<T0, TN, i0, i1 - i0>(a, lds, buf2, s + 2);
                else dma_chunks<T0N, TNN, i0, i1 - i0>(a, lds, buf2, s + 2 - NT);
              }
            }
          } else if constexpr (u + 1 == UNITS) {
            constexpr int NMF = two ? 12 : 6, PER = (NCH + NMF - 1) / NMF;
            if constexpr (j * PER < NCH) {
              constexpr int n = (j + 1) * PER <= NCH ? PER : NCH - j * PER;
              if constexpr (s + 2 < NT) dma_chunks<T0, TN, j * PER, n>(a, lds, buf, s + 2);
              else dma_chunks<T0N, TNN, j * PER, n>(a, lds, buf, s + 2 - NT);
            }
          }
#endif
        };
        bf16x8 anext[6];
        auto prefetch = [&](auto ic) {  // part i of the next unit's operands
          if constexpr (u + 1 < UNITS) read_a_part(sb, std::integral_constant<int, (u + 1) / PAIRS>{}, std::integral_constant<int, (u + 1) % PAIRS>{}, ic, anext);
          else if constexpr (s + 1 < NT) read_a_part(sbn, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, ic, anext);
        };
        __builtin_amdgcn_sched_barrier(0);
        f32x16 g0 = gout[l0], g1;
        if constexpr (two) g1 = gout[l1];
        static_for<6>([&](auto tc) {  // (term, operand) in issue order: Pl dh | Pm dm, Pm dh | Ph dl, Ph dm, Ph dh
          constexpr int term = decltype(tc)::value;
          constexpr int ai = term == 0 ? 0 : (term <= 2 ? 1 : 2);
          const bf16x8& bp = (term == 0 || term == 2 || term == 5) ? bb.h : ((term == 1 || term == 4) ? bb.m : bb.l);
#ifdef EBM_ABL_NOMFMA
          asm volatile("" :: "v"(acur[ai]), "v"(bp));
#else
          g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[ai], bp, g0, 0, 0, 0);
#endif
          if constexpr (term < 3) prefetch(tc);
          request(std::integral_constant<int, (two ? 2 : 1) * term>{});
          slot(std::integral_constant<int, o0 + (two ? 2 : 1) * term>{});
          if constexpr (two) {
#ifdef EBM_ABL_NOMFMA
            asm volatile("" :: "v"(acur[3 + ai]), "v"(bp));
#else
            g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[3 + ai], bp, g1, 0, 0, 0);
#endif
            request(std::integral_constant<int, 2 * term + 1>{});
            slot(std::integral_constant<int, o0 + 2 * term + 1>{});
          }
        });
        gout[l0] = g0;
        if constexpr (two) gout[l1] = g1;
        if constexpr (u + 1 < UNITS || s + 1 < NT) {
#pragma unroll
          for (int i = 0; i < 6; ++i) acur[i] = anext[i];
        }
      });
      if constexpr (s + 1 < NT) b0 = jb0n.tri();
      ++gstage;
      gbuf = bufn;
    });
    EBM_HSTAMP(3);
    float acc = 0.0f;
    if (want_e) {  // (wave-uniform: one copy of the code serves the trajectory's interior and its last step)
      static_for<TN * 4>([&](auto ic) {
        constexpr int t = T0 + (decltype(ic)::value >> 2), tl = decltype(ic)::value >> 2, q = decltype(ic)::value & 3;
        const f32x4 mq = *reinterpret_cast<const f32x4*>(mus + 32 * t + 8 * q + 4 * hs);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_fmaf(x.t[t][4 * q + i] - mq[i], gout[tl][4 * q + i], acc);
      });
      acc += __shfl_xor(acc, 32);
    }
    asm volatile("" : "+v"(acc));
    EBM_HSTAMP(4);
#ifdef EBM_PHASE_TIMES
    if (gstage == NT * 20 && blockIdx.x == 0 && threadIdx.x == 0) {  // (ten evaluations in: the first pass's "between" is the prologue)
      for (int i = 0; i < 6; ++i) ebm_hmc_phase_log[i] = ph_acc_[i];
      ebm_hmc_phase_log[6] = (unsigned long long)(gstage / NT);
    }
#endif
    return 0.5f * acc;
  }
};

}  // namespace
}  // namespace ebm
