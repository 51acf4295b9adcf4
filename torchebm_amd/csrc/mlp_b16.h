// The wide-MLP contractions on the bf16 matrix pipe at fp32 accuracy (MODE 2 of mlp_wide_body.h).
//
// The exact-f32 MFMA (v_mfma_f32_32x32x2_f32) runs at the f32 vector rate on the VALU's own lanes: an evaluation at H = 128
// is 640 instructions of 64 cycles that never overlap the SiLU / Philox work.  v_mfma_f32_32x32x16_bf16 is 16x that rate on
// a separate pipe; with both operands split three ways (gauss_bf16x3.h: v = hi + mid + lo, six products of total order <= 2,
// smallest first, fp32 accumulation) a contraction costs 6/16 of the f32 matrix time and leaves the VALU free.
//
// ONE LDS image per weight matrix serves both walks.  The forward walks (W1 x, W2 h1) want lane (m, h) to hold row 32 it + m
// at the eight columns the C/D layout of the previous result gives K-block kb of lane-half h -- 16 kb + 4 h + {0..3} and
// + 8 of that -- and the transposed walks (W2^T d2, W1^T d1) want COLUMN 32 it + m at eight such rows.  A second, transposed
// copy of W2 (96 KB of bf16 triples at H = 128) does not fit.  Instead the image is row-major bf16 made of 8-byte atoms
// (row, four consecutive columns), stored so that
//   * the two atoms a forward lane needs are adjacent: one ds_read_b128 per operand;
//   * the transposed operand is gathered by ds_read_b64_tr_b16 (gfx950's transpose read: sixteen lanes hand in the sixteen
//     atoms of a [4 rows] x [16 columns] block, lane c gets column c of it): two per operand;
//   * both are bank-conflict free: the 16-byte unit index inside a row is XORed with a bit rotation of the row index that
//     sends the row bits a forward lane group varies (m & 15) to all unit bits, and the two row bits a transposed gather
//     varies (the four rows of a block) to the unit bits the gather does not vary itself (its output tile's).
//     Checked against the bank model of MI355X_MICROARCH.md (b128: four groups of 16 lanes over 64 banks; tr_b16: two
//     groups of 32) for C = 32 / 64 / 128: 4 and 2 LDS cycles per instruction, the conflict-free minimum.
#pragma once
#include "ebm_common.h"
#include "gauss_bf16x3.h"

namespace ebm {
namespace mlpb16 {

using gauss3::bf16x8;
using gauss3::f32x16;
using gauss3::f32x8;
using gauss3::static_for;
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) char* lds_bytes;
typedef __attribute__((address_space(3))) bf16x4* lds_bf16x4;
typedef __attribute__((address_space(3))) bf16x8* lds_bf16x8;

// One split of an [R][C] matrix: rows of 2 C bytes, C in {32, 64, 128}.
template <int C>
struct Img {
  static_assert(C == 32 || C == 64 || C == 128, "image width");
  static constexpr int RB = 2 * C;                               // bytes per row
  static constexpr int NB = C == 128 ? 4 : (C == 64 ? 3 : 2);     // bits of the 16-byte unit index inside a row
  static constexpr int LB = 4 - NB;                               // log2(rows per 256-byte bank row)
  static constexpr int AB = 2 - LB;                               // block-row bits that do not already select a bank-row part
  __host__ __device__ static constexpr uint32_t swz(uint32_t row) {
    const uint32_t v = (row >> LB) & ((1u << NB) - 1u);
    return 16u * (((v & ((1u << AB) - 1u)) << (NB - AB)) | (v >> AB));
  }
  // byte offset of the atom (row, columns 4 cq .. 4 cq + 3); cq = 4 kb + 2 s + h': K-block, first / second quad, lane half
  __host__ __device__ static constexpr uint32_t atom(uint32_t row, uint32_t cq) {
    const uint32_t colpos = 16u * (2u * (cq >> 2) + (cq & 1u)) + 8u * ((cq >> 1) & 1u);
    return row * (uint32_t)RB + (colpos ^ swz(row));
  }
};

__host__ __device__ constexpr size_t image_bytes(int rows, int cols) { return (size_t)3 * rows * cols * 2; }

// ---- MODE 3 (round 4): W1 at H = 128 and dim > 64.  Both images are 96 KB -- 192 KB against the CU's 160.  W2 stays
// resident; W1 is kept in GLOBAL memory as a PRE-SPLIT image (ebm_mlp_w1_image_f32 builds it: the caller's buffer, handed
// over in ebm_energy_t.aux) made of four SLABS of 32 hidden rows: slab s = the three split images [32][128] of rows
// 32 s .. 32 s + 31, each exactly what stage_image<32, 128> would write (24 KB).  A slab is what ONE output tile of the
// forward walk (W1 x: all eight K-blocks of rows 32 s ..) and ONE pair of K-blocks of the transposed walk (W1^T d1) read, so
// both walks run slab by slab out of two 24 KB LDS buffers, filled by LDS-direct loads (global_load_lds_dwordx4: no
// registers, no ds_write pass) one slab ahead of the MFMAs -- 48 of them per slab and wave to land in.
constexpr int kSlabRows = 32, kSlabCols = 128;
constexpr uint32_t kSlabBytes = 3u * kSlabRows * kSlabCols * 2u;  // 24 576
// All threads: request slab `s` of the global image into the LDS buffer at byte address `dst`.  One wave-instruction moves
// 1 KiB -- lane l's 16 bytes land at M0 + 16 l, so the copy is linear -- 24 pieces, six per wave of a 256-thread workgroup.
// Written as assembly ON PURPOSE: after the builtin the compiler assumes every later LDS read may alias the transfer and
// puts s_waitcnt vmcnt(0) in front of it, which serialises exactly the overlap this is for; slab_wait() is the one wait.
// (The buffers sit below 64 KiB -- mlp_wide_setup.inc -- so M0 holds a plain 16-bit LDS address.)
__device__ __forceinline__ void slab_request(const char* image, int s, uint32_t dst, int n_threads) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63, waves = n_threads >> 6;
  const char* src = image + (size_t)s * kSlabBytes;
  for (int piece = wave; piece < (int)(kSlabBytes / 1024u); piece += waves) {
    const uint32_t voff = (uint32_t)(piece * 1024 + lane * 16), base = dst + (uint32_t)piece * 1024u;
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(src), "s"(base) : "memory");
  }
}
// Before a slab is read: this wave's pieces have landed, and so have everybody else's (the barrier); every wave has also
// finished with the buffer used before, which the next request overwrites.
__device__ __forceinline__ void slab_wait() {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
}

// All threads of the workgroup: W[rows_real][cols_real] (fp32, row-major, global) -> three split images of [R][C] in LDS
// (split s at img + s R 2 C), zero beyond the real extent.
template <int R, int C>
__device__ __forceinline__ void stage_image(const float* __restrict__ w, int rows_real, int cols_real, lds_bytes img, int n_threads) {
  constexpr int QPR = C / 4;
  constexpr uint32_t SPLIT = (uint32_t)R * 2u * C;
  const bool quads = (cols_real & 3) == 0 && (reinterpret_cast<uintptr_t>(w) & 15) == 0;
  for (int i = threadIdx.x; i < R * QPR; i += n_threads) {
    const int row = i / QPR, cq = i - row * QPR;
    float v[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    if (row < rows_real) {
      if (quads && 4 * cq + 3 < cols_real) {
        const float4 q = *reinterpret_cast<const float4*>(w + (size_t)row * cols_real + 4 * cq);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (4 * cq + e < cols_real) v[e] = w[(size_t)row * cols_real + 4 * cq + e];
      }
    }
    bf16x4 hi, mid, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      __bf16 a, b, c;
      gauss3::split3(v[e], a, b, c);
      hi[e] = a; mid[e] = b; lo[e] = c;
    }
    const uint32_t off = Img<C>::atom((uint32_t)row, (uint32_t)cq);
    *(lds_bf16x4)(img + off) = hi;
    *(lds_bf16x4)(img + SPLIT + off) = mid;
    *(lds_bf16x4)(img + 2u * SPLIT + off) = lo;
  }
}

// ---------------------------------------------------------------------------------------------------------------------------
// The pipelined contraction.  ONE wave per SIMD (the images leave room for one workgroup per CU, the live tiles for one wave
// per SIMD) issues one vector instruction per 5.0 cycles (8.0 for a transcendental; profiles/r06_mfma_valu_overlap.txt, the pinned
// stream), and FIVE of them hide behind a 32-cycle MFMA -- the sixth costs its issue time, a packed-f32 one 20 cycles more: per
// gap max(32 + 0.5 n, 10 + 5 n).  An evaluation is ~400 - 500 MFMAs next to ~2 000 - 2 500 other instructions, 4 - 6 per MFMA, so what
// it costs is its instruction COUNT -- provided MFMAs and the rest alternate, <= 5 per gap, nothing packed (round 6: see EBM_PIN and the
// micro-slots below for what kept them from alternating).
// Hence: the epilogue of the PREVIOUS contraction's tile j + 1 (which yields K-blocks 2 j + 2, 2 j + 3 of this one) is issued
// in slices behind the MFMAs of this contraction's K-blocks 2 j, 2 j + 1 (`fill(ordinal)`, called once behind every MFMA
// and fenced there, the MFMA from its fill as well); pair arithmetic behind MFMAs is element-wise, in MFMA-free stretches packed
// (pmul / pfma<PK>); an operand address is one lane register + an immediate; a group's operands are awaited once.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// A K-block's three pieces as four packed pairs each (pair p = elements 2 p, 2 p + 1): an epilogue completes it pair by pair.
struct Split8p {
  u32x4 h, m, l;
};
__device__ __forceinline__ uint32_t cvt_pair(f32x2 v) { return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2)); }
__device__ __forceinline__ f32x2 widen_pair(uint32_t packed) {
  u32x2 w = {packed, packed};
  w = (w << (u32x2){16u, 0u}) & (u32x2){0xffffffffu, 0xffff0000u};
  return __builtin_bit_cast(f32x2, w);
}
// (round 6) EBM_PIN: an epilogue is issued slot by slot behind MFMAs, and the slots of the tile that hides in a contraction's
// tile-major tail sit in front of a branch (EBM_BLOCK_CUT / `eval_energy_only`) whose other successor does not use their results --
// LLVM's code sinking then moved the whole epilogue (~300 instructions) out of the 36 - 48 MFMA gaps it was written into and
// into the successor block, where it ran with the matrix pipe idle (scripts/isa_gaps.py: 96 empty gaps per evaluation).  A volatile
// empty asm on the values a slot chain ENDS in keeps the chain in the block it is written in (inside the block the MFMA / fill
// fences keep the order).  One asm per tile epilogue, not per slot: the hazard recogniser puts an s_nop behind every asm.
#define EBM_PIN(v) asm volatile("" : "+v"(v))
// Pair arithmetic in two spellings (round 6; profiles/r06_mfma_valu_overlap.txt).  PK = true: <2 x float> operations, which select
// v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32 -- half the issue slots of a lone wave, the form for MFMA-FREE stretches.  PK = false: the
// same IEEE operations element by element (bit-identical results) -- the form for slots issued BEHIND an MFMA, where the first packed
// instruction of a gap stalls the wave ~20 cycles (one v_pk_fma_f32 per gap: 54 cycles per MFMA; the two v_fma_f32 it replaces: 35).
// The MLP units are compiled with -fno-slp-vectorize so that the element-wise spelling is not packed again (csrc/Makefile).
template <bool PK> __device__ __forceinline__ f32x2 pmul(f32x2 a, f32x2 b) {
  if constexpr (PK) return a * b; else return (f32x2){a.x * b.x, a.y * b.y};
}
template <bool PK> __device__ __forceinline__ f32x2 pmul(f32x2 a, float b) {
  if constexpr (PK) return a * b; else return (f32x2){a.x * b, a.y * b};
}
template <bool PK> __device__ __forceinline__ f32x2 padd(f32x2 a, float b) {
  if constexpr (PK) return a + b; else return (f32x2){a.x + b, a.y + b};
}
template <bool PK> __device__ __forceinline__ f32x2 psub(f32x2 a, f32x2 b) {
  if constexpr (PK) return a - b; else return (f32x2){a.x - b.x, a.y - b.y};
}
template <bool PK> __device__ __forceinline__ f32x2 prsub(float a, f32x2 b) {  // a - b
  if constexpr (PK) return a - b; else return (f32x2){a - b.x, a - b.y};
}
template <bool PK> __device__ __forceinline__ f32x2 pfma(f32x2 a, f32x2 b, f32x2 c) {
  if constexpr (PK) return __builtin_elementwise_fma(a, b, c);
  else return (f32x2){__builtin_fmaf(a.x, b.x, c.x), __builtin_fmaf(a.y, b.y, c.y)};
}
// stage A of a pair split: hi piece + residual; stage B: mid piece + residual; stage C: lo piece (the residual of B has at
// most eight significant bits: the conversion is exact)
template <int P, bool PK = true>
__device__ __forceinline__ void pair_split_a(Split8p& s, f32x2& r, f32x2 v) {
  const uint32_t ph = cvt_pair(v);
  s.h[P] = ph;
  r = psub<PK>(v, widen_pair(ph));
}
template <int P, bool PK = true>
__device__ __forceinline__ void pair_split_b(Split8p& s, f32x2& r) {
  const uint32_t pm = cvt_pair(r);
  s.m[P] = pm;
  r = psub<PK>(r, widen_pair(pm));
}
// The same split in micro-steps of two or three instructions (round 6: an epilogue is dealt out behind MFMAs ~5 instructions per
// gap -- five hide, the sixth costs its issue time -- so its units must be finer than a stage): `piece`: the bf16 pair of v and its
// widened copy; `resid`: what is left.
template <int P, int STAGE>  // STAGE 0: hi, 1: mid
__device__ __forceinline__ void pair_split_piece(Split8p& s, f32x2& wide, f32x2 v) {
  const uint32_t pc = cvt_pair(v);
  if constexpr (STAGE == 0) s.h[P] = pc; else s.m[P] = pc;
  wide = widen_pair(pc);
}
template <bool PK>
__device__ __forceinline__ void pair_split_resid(f32x2& r, f32x2 v, f32x2 wide) {
  r = psub<PK>(v, wide);
}
template <int P>
__device__ __forceinline__ void pair_split_c(Split8p& s, f32x2 r) {
  s.l[P] = cvt_pair(r);
}
template <int R0>
__device__ __forceinline__ f32x2 pair_of(const f32x16& tile) {
  return __builtin_shufflevector(tile, tile, R0, R0 + 1);
}

struct NoFill {
  template <class O>
  __device__ __forceinline__ void operator()(O) const {}
};

// A never-taken branch on an opaque scalar: it ends the basic block.  The evaluation is otherwise ONE block of ~4 000
// instructions, over which the scheduler stretches live ranges until a 512-register wave spills; cut into its four phases the
// same code needs ~380 registers (scripts/kernel_regs.py).  Two scalar instructions per cut.
#define EBM_BLOCK_CUT()                         \
  do {                                          \
    int never_ = 0;                             \
    asm volatile("" : "+s"(never_));            \
    if (never_ != 0) __builtin_trap();          \
  } while (0)

#ifdef EBM_ABL_NOWAIT  /* timing ablation only (scripts/ab_build.sh ... -DEBM_ABL_NOWAIT): wrong results */
#define EBM_WAIT_LDS() do {} while (0)
#else
#define EBM_WAIT_LDS() __builtin_amdgcn_s_waitcnt(0xc07f) /* lgkmcnt(0), vmcnt / expcnt untouched */
#endif

// The A operands of the two walks over an image of width C with R rows (split stride R 2 C).  Every address is ONE lane
// register (per K-block of the forward walk; per (output tile, half) of the transposed walk) plus an immediate; the one
// image whose far split lies beyond the 64 KiB an LDS offset reaches (W2 at H = 128) gets a second register for it.
template <int R, int C>
struct Walk {
  static constexpr uint32_t RB = Img<C>::RB, SPLIT = (uint32_t)R * RB;
  uint32_t base;       // LDS byte address of split 0
  uint32_t fwd_lane;   // forward: row m
  uint32_t fwd_x;      // ... its unit h, swizzled
  uint32_t bwd_lane;   // transposed: row 4 h + (i >> 2) of a 16-row K-block
  uint32_t bwd_x;      // ... columns 16 g + 4 (i & 3) .. + 3, swizzled by that row
  __device__ __forceinline__ Walk(lds_bytes img, int lane) {
    base = (uint32_t)(uintptr_t)img;
    const uint32_t m = lane & 31, h = lane >> 5, i = lane & 15, g = (lane >> 4) & 1;
    fwd_lane = m * RB;
    fwd_x = (16u * h) ^ Img<C>::swz(m);
    const uint32_t row = 4u * h + (i >> 2);
    bwd_lane = row * RB;
    bwd_x = (32u * g + 16u * (i & 1u) + 8u * ((i >> 1) & 1u)) ^ Img<C>::swz(row);
  }
  // The images never change after staging, so every operand load is invariant over the step loop: left visible, the
  // compiler hoists them all (a whole image per wave) and spills.  An opaque copy per evaluation keeps them at their MFMAs.
  __device__ __forceinline__ Walk opaque() const {
    Walk w = *this;
    asm volatile("" : "+v"(w.fwd_lane), "+v"(w.bwd_lane));
    return w;
  }
  struct Addr {
    uint32_t a, far;  // far = a + 64 KiB (only formed where an offset can exceed the field)
  };
  template <uint32_t OFF>
  static __device__ __forceinline__ lds_bytes at(const Addr& b) {
    if constexpr (OFF < 65536u) return (lds_bytes)(uintptr_t)(b.a + OFF);
    else return (lds_bytes)(uintptr_t)(b.far + (OFF - 65536u));
  }
  static constexpr bool FAR_F = 2u * SPLIT + (R / 32 - 1) * 32u * RB >= 65536u;
  static constexpr bool FAR_B = 2u * SPLIT + (R - 8) * RB >= 65536u;
  // forward, K-block kb: one register for every (split, output tile)
  template <int KBI>
  __device__ __forceinline__ Addr fwd_kb() const {
    Addr b;
    b.a = base + fwd_lane + (fwd_x ^ (32u * KBI));
    b.far = b.a;
    if constexpr (FAR_F) b.far = b.a + 65536u;
    asm volatile("" : "+v"(b.a), "+v"(b.far));  // one value each, not re-derived per load
    return b;
  }
  template <int SP, int IT>
  static __device__ __forceinline__ bf16x8 fwd_load(const Addr& b) {
    return *(lds_bf16x8)at<(uint32_t)SP * SPLIT + (uint32_t)IT * 32u * RB>(b);
  }
  // transposed, output tile it, half s (rows + 8 s of a K-block): one register for every (split, K-block)
  template <int IT, int S>
  __device__ __forceinline__ Addr bwd_tile() const {
    constexpr uint32_t c = (64u * IT) ^ (S ? Img<C>::swz(8u) : 0u);  // the swizzle is XOR-linear in the row bits
    Addr b;
    b.a = base + bwd_lane + (bwd_x ^ c) + (S ? 8u * RB : 0u);
    b.far = b.a;
    if constexpr (FAR_B) b.far = b.a + 65536u;
    asm volatile("" : "+v"(b.a), "+v"(b.far));
    return b;
  }
  template <int SP, int KBI>
  static __device__ __forceinline__ bf16x8 bwd_load(const Addr& b0, const Addr& b1) {
    constexpr uint32_t off = (uint32_t)SP * SPLIT + 16u * KBI * RB;
    const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)at<off>(b0));
    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds_bf16x4)at<off>(b1));
    return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

// out[it] (initialised by the caller: bias tiles or zero) += A_it B over K-blocks 0 .. KB - 1.
//   TR: the transposed walk (A = image^T);  bs(kb) -> const Split8p& of K-block kb (complete before ordinal 6 NT kb);
//   fill(ordinal): ordinals 0 .. 6 NT KB - 1 in issue order.
// K-block outermost: groups (kb, A split) in the order lo | mid | hi with 1 | 2 | 3 terms each (smallest products first),
// term-major inside a group (NT independent MFMAs between dependent ones); NT == 1 alternates two accumulators.  A group's
// operands are requested in front of the group before (a whole group of MFMAs to land in) and awaited once.
// TAIL: the last two K-blocks are issued TILE-major instead (the 12 MFMAs of tile 0, then tile 1's ...): tile 0 is final
// 12 (NT - 1) MFMAs before the contraction ends and the caller's fill runs tile 0's epilogue behind those -- otherwise every
// tile is final at the same moment and the first epilogue of the next stage has no MFMA to hide behind.
template <int NT, int KB, bool TAIL, bool TR, class W, class Bs, class Fill>
__device__ __forceinline__ void contract_pipe(f32x16 (&out)[NT], const W& w, Bs bs, Fill fill) {
  static_assert(!TAIL || NT > 1, "a tail needs a second tile to hide behind");
  constexpr int KBH = TAIL ? KB - 2 : KB;  // K-blocks of the K-block-major head
  constexpr int G = 3 * KBH;
  constexpr int SETS = NT == 1 ? 2 : 1;
  typedef typename W::Addr Addr;
  f32x16 extra;
#pragma unroll
  for (int r = 0; r < 16; ++r) extra[r] = 0.0f;
  bf16x8 pa[2][NT];
  bf16x8 pt[2][6];  // TAIL: the six operands (two K-blocks x lo, mid, hi) of one tile, requested one tile ahead
  Addr bt[TR ? NT : 1][2];  // transposed walk: the (tile, half) registers
  if constexpr (TR)
    static_for<NT>([&](auto itc) __attribute__((always_inline)) {
      constexpr int it = decltype(itc)::value;
      bt[it][0] = w.template bwd_tile<it, 0>();
      bt[it][1] = w.template bwd_tile<it, 1>();
    });
  Addr fk;  // forward walk: the register of the K-block being requested
  // operands of group (kb, grp) for every tile
  const auto load_group = [&](auto kbc, auto grpc, auto bufc) __attribute__((always_inline)) {
    constexpr int kb = decltype(kbc)::value, grp = decltype(grpc)::value, buf = decltype(bufc)::value;
#ifdef EBM_ABL_NOLOAD  /* timing ablation only: the operands are whatever the registers hold */
    if constexpr (kb > 0 || grp > 0) return;
#endif
    if constexpr (!TR && grp == 0) fk = w.template fwd_kb<kb>();
    static_for<NT>([&](auto itc) __attribute__((always_inline)) {
      constexpr int it = decltype(itc)::value;
      if constexpr (TR) pa[buf][it] = W::template bwd_load<2 - grp, kb>(bt[it][0], bt[it][1]);
      else pa[buf][it] = W::template fwd_load<2 - grp, it>(fk);
    });
  };
  Addr ft[2];  // TAIL, forward walk: the registers of the two K-blocks
  const auto load_tail = [&](auto itc) __attribute__((always_inline)) {
    constexpr int it = decltype(itc)::value;
#ifdef EBM_ABL_NOLOAD
    if constexpr (it > 0) return;
#endif
    static_for<6>([&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;  // K-block q / 3 of the tail, group q % 3 (0: lo)
      if constexpr (TR) pt[it & 1][q] = W::template bwd_load<2 - q % 3, KBH + q / 3>(bt[it][0], bt[it][1]);
      else pt[it & 1][q] = W::template fwd_load<2 - q % 3, it>(ft[q / 3]);
    });
  };
  if constexpr (TAIL && !TR) {
    ft[0] = w.template fwd_kb<KBH>();
    ft[1] = w.template fwd_kb<KBH + 1>();
  }
  if constexpr (G > 0) load_group(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
  else load_tail(std::integral_constant<int, 0>{});
  static_for<G>([&](auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value, kb = g / 3, grp = g % 3;
    constexpr int n_terms = grp + 1, first = grp * (grp + 1) / 2;
    const Split8p& b = bs(std::integral_constant<int, kb>{});
    EBM_WAIT_LDS();
    __builtin_amdgcn_sched_barrier(0);
    // the next group's operands: a whole group of MFMAs (4 / 8 / 12 NT / 4) to land in
    if constexpr (g + 1 < G)
      load_group(std::integral_constant<int, (g + 1) / 3>{}, std::integral_constant<int, (g + 1) % 3>{}, std::integral_constant<int, (g + 1) & 1>{});
    else if constexpr (TAIL)
      load_tail(std::integral_constant<int, 0>{});
    __builtin_amdgcn_sched_barrier(0);
    static_for<n_terms * NT>([&](auto oc) __attribute__((always_inline)) {
      constexpr int o = decltype(oc)::value, tg = o / NT, it = o % NT;
      constexpr int term = first + tg;  // 0: Al dh | 1: Am dm, 2: Am dh | 3: Ah dl, 4: Ah dm, 5: Ah dh
      const bf16x8 db = __builtin_bit_cast(bf16x8, (term == 0 || term == 2 || term == 5) ? b.h : ((term == 1 || term == 4) ? b.m : b.l));
      if constexpr (SETS == 2 && (term & 1)) extra = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[g & 1][it], db, extra, 0, 0, 0);
      else out[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[g & 1][it], db, out[it], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // (round 6: the MFMA FIRST -- left in one region with its fill, a group's first MFMA sank below it)
      fill(std::integral_constant<int, (kb * 6 + term) * NT + it>{});
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  if constexpr (TAIL) {
    static_for<NT>([&](auto itc) __attribute__((always_inline)) {
      constexpr int it = decltype(itc)::value;
      EBM_WAIT_LDS();
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (it + 1 < NT) {
        load_tail(std::integral_constant<int, it + 1>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      static_for<12>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value, kbl = q / 6, term = q % 6, grp = term == 0 ? 0 : (term < 3 ? 1 : 2);
        const Split8p& b = bs(std::integral_constant<int, KBH + kbl>{});
        const bf16x8 db = __builtin_bit_cast(bf16x8, (term == 0 || term == 2 || term == 5) ? b.h : ((term == 1 || term == 4) ? b.m : b.l));
        out[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pt[it & 1][3 * kbl + grp], db, out[it], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        fill(std::integral_constant<int, KBH * 6 * NT + it * 12 + q>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  if constexpr (SETS == 2) out[0] += extra;
}

// Slots of a tile epilogue behind the MFMAs of one block (two K-blocks = NMB MFMAs): MFMA ob of the block runs slots
// [NS ob / NMB, NS (ob + 1) / NMB) of the NS the epilogue of a 16-register tile is cut into (round 6: micro-slots of 2 - 3
// instructions -- 12 / 14 / 6 per pair for the three epilogues --, so that every gap gets its ~5).
template <int NMB, int OB, int NS = 48, class Slot>
__device__ __forceinline__ void run_slots(Slot slot) {
  constexpr int s0 = NS * OB / NMB, s1 = NS * (OB + 1) / NMB;
  static_for<s1 - s0>([&](auto k) __attribute__((always_inline)) { slot(std::integral_constant<int, s0 + decltype(k)::value>{}); });
}

}  // namespace mlpb16
}  // namespace ebm
