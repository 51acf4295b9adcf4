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

// three bf16 pieces of the eight K values a lane supplies to one K-block
struct Split8 {
  bf16x8 h, m, l;
};
// Written on the packed conversions: one v_cvt_pk_bf16_f32 per pair and piece, the pair widened back to fp32 with a shift
// and a mask (left to __builtin_convertvector the compiler re-converts every element on its own to widen it: 60 instead
// of 44 instructions per K-block).
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  const f32x2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ Split8 split8(const f32x8& d) {
  uint32_t ph[4], pm[4], pl[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float a = d[2 * i], b = d[2 * i + 1];
    ph[i] = cvt_pk_bf16(a, b);
    const float ra = a - __builtin_bit_cast(float, ph[i] << 16), rb = b - __builtin_bit_cast(float, ph[i] & 0xffff0000u);
    pm[i] = cvt_pk_bf16(ra, rb);
    const float sa = ra - __builtin_bit_cast(float, pm[i] << 16), sb = rb - __builtin_bit_cast(float, pm[i] & 0xffff0000u);
    pl[i] = cvt_pk_bf16(sa, sb);
  }
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  Split8 s;
  s.h = __builtin_bit_cast(bf16x8, (u32x4){ph[0], ph[1], ph[2], ph[3]});
  s.m = __builtin_bit_cast(bf16x8, (u32x4){pm[0], pm[1], pm[2], pm[3]});
  s.l = __builtin_bit_cast(bf16x8, (u32x4){pl[0], pl[1], pl[2], pl[3]});
  return s;
}

// out[it] += A_it B over KB K-blocks of 16, K-block outermost: the eight B values of a K-block (breg(kb, j): element j of
// this lane for K-block kb, in the C/D layout of the 32-row tile kb >> 1: register 8 (kb & 1) + j) are split once and meet
// all NT output tiles; per K-block the six terms run smallest first, grouped by the split of A they read (lo: dh | mid: dm,
// dh | hi: dl, dm, dh), term-major inside a group so that NT independent MFMAs sit between dependent ones.
// lda(split, it, kb) -> this lane's A operand.
template <int NT, int KB, class Lda, class Breg>
__device__ __forceinline__ void contract(f32x16 (&out)[NT], Lda lda, Breg breg) {
  constexpr int SETS = NT == 1 ? 2 : 1;  // one tile: two accumulators, so that consecutive MFMAs never wait on each other
  f32x16 extra[SETS == 2 ? 1 : 1];
#pragma unroll
  for (int r = 0; r < 16; ++r) extra[0][r] = 0.0f;
  static_for<KB>([&](auto kbc) __attribute__((always_inline)) {
    f32x8 d;
    static_for<8>([&](auto jc) __attribute__((always_inline)) { d[decltype(jc)::value] = breg(kbc, jc); });
    const Split8 b = split8(d);
    static_for<3>([&](auto gc) __attribute__((always_inline)) {
      constexpr int grp = decltype(gc)::value;  // 0: A lo, 1: A mid, 2: A hi
      constexpr int n_terms = grp + 1, first = grp * (grp + 1) / 2;
      bf16x8 pa[NT];
      static_for<NT>([&](auto itc) __attribute__((always_inline)) {
        pa[decltype(itc)::value] = lda(std::integral_constant<int, 2 - grp>{}, itc, kbc);
      });
      static_for<n_terms * NT>([&](auto oc) __attribute__((always_inline)) {
        constexpr int o = decltype(oc)::value, tg = o / NT, it = o % NT;
        constexpr int term = first + tg;  // 0: Al dh | 1: Am dm, 2: Am dh | 3: Ah dl, 4: Ah dm, 5: Ah dh
        const bf16x8& db = (term == 0 || term == 2 || term == 5) ? b.h : ((term == 1 || term == 4) ? b.m : b.l);
        if constexpr (SETS == 2 && (term & 1)) extra[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[it], db, extra[0], 0, 0, 0);
        else out[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[it], db, out[it], 0, 0, 0);
      });
    });
  });
  if constexpr (SETS == 2) out[0] += extra[0];
}

// ---------------------------------------------------------------------------------------------------------------------------
// The pipelined form.  One wave per SIMD issues about one instruction per four cycles whatever its kind, and a split-operand
// evaluation is ~480 MFMAs of 32 cycles next to ~3500 VALU instructions: run one after the other (contract() above, then
// the SiLU epilogue, then the next contract()) the matrix pipe idles through every epilogue and the VALU through every
// contraction.  Here the epilogue of the PREVIOUS contraction's tile j + 1 (which yields K-blocks 2 j + 2, 2 j + 3 of this one)
// is issued in slices behind the MFMAs of this contraction's K-blocks 2 j, 2 j + 1: `fill(ordinal)` is called once behind
// every MFMA and fenced there (sched_barrier), the A operands of the next group are requested one group ahead.
//
// Split8 pieces are addressed as four packed pairs each (pair p = elements 2 p, 2 p + 1), so that an epilogue can complete
// a K-block pair by pair.
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
struct Split8p {
  u32x4 h, m, l;
};
// stage A of a pair split: hi piece + residuals; stage B: mid piece + residuals; stage C: lo piece.  Pairs are f32x2 so that
// the arithmetic is the packed form (v_pk_*): a lone wave per SIMD issues about one instruction per six cycles whatever the
// instruction is, so the instruction COUNT is what an epilogue costs.
struct PairSplit {
  f32x2 r;
};
__device__ __forceinline__ f32x2 widen_pair(uint32_t packed) {
  f32x2 w;
  w.x = __builtin_bit_cast(float, packed << 16);
  w.y = __builtin_bit_cast(float, packed & 0xffff0000u);
  return w;
}
template <int P>
__device__ __forceinline__ void pair_split_a(Split8p& s, PairSplit& t, f32x2 v) {
  const uint32_t ph = cvt_pk_bf16(v.x, v.y);
  s.h[P] = ph;
  t.r = v - widen_pair(ph);
}
template <int P>
__device__ __forceinline__ void pair_split_b(Split8p& s, PairSplit& t) {
  const uint32_t pm = cvt_pk_bf16(t.r.x, t.r.y);
  s.m[P] = pm;
  t.r = t.r - widen_pair(pm);
}
template <int P>
__device__ __forceinline__ void pair_split_c(Split8p& s, const PairSplit& t) {
  s.l[P] = cvt_pk_bf16(t.r.x, t.r.y);
}

struct NoFill {
  template <class O>
  __device__ __forceinline__ void operator()(O) const {}
};

// out[it] (initialised by the caller: bias tiles or zero) += A_it B over K-blocks 0 .. KB - 1.
//   lda(split, it, kb) -> this lane's A operand;  bs(kb) -> const Split8p& of K-block kb (complete before ordinal 6 NT kb);
//   fill(ordinal): ordinals 0 .. 6 NT KB - 1 in issue order.
// K-block outermost: groups (kb, A split) in the order lo | mid | hi with 1 | 2 | 3 terms each, term-major inside a group (NT
// independent MFMAs between dependent ones); NT == 1 alternates two accumulators.
// TAIL: the last two K-blocks are issued TILE-major instead (the 12 MFMAs of tile 0, then tile 1's ...): tile 0 is final
// 12 (NT - 1) MFMAs before the contraction ends, and the caller's fill runs tile 0's epilogue behind those -- otherwise every
// tile is final at the same moment and the first epilogue of the next stage has no MFMA to hide behind.
template <int NT, int KB, bool TAIL, class Lda, class Bs, class Fill>
__device__ __forceinline__ void contract_pipe(f32x16 (&out)[NT], Lda lda, Bs bs, Fill fill) {
  static_assert(!TAIL || NT > 1, "a tail needs a second tile to hide behind");
  constexpr int KBH = TAIL ? KB - 2 : KB;  // K-blocks of the K-block-major head
  constexpr int G = 3 * KBH;
  constexpr int SETS = NT == 1 ? 2 : 1;
  f32x16 extra;
#pragma unroll
  for (int r = 0; r < 16; ++r) extra[r] = 0.0f;
  bf16x8 pa[2][NT];
  bf16x8 pt[2][6];  // TAIL: the six operands (two K-blocks x lo, mid, hi) of one tile, requested one tile ahead
  const auto load_tail = [&](auto itc) __attribute__((always_inline)) {
    constexpr int it = decltype(itc)::value;
    static_for<6>([&](auto qc) __attribute__((always_inline)) {
      constexpr int q = decltype(qc)::value;  // K-block q / 3 of the tail, group q % 3 (0: lo)
      pt[it & 1][q] = lda(std::integral_constant<int, 2 - q % 3>{}, itc, std::integral_constant<int, KBH + q / 3>{});
    });
  };
  if constexpr (G > 0) {
    static_for<NT>([&](auto itc) __attribute__((always_inline)) {
      pa[0][decltype(itc)::value] = lda(std::integral_constant<int, 2>{}, itc, std::integral_constant<int, 0>{});
    });
  } else {
    load_tail(std::integral_constant<int, 0>{});
  }
  static_for<G>([&](auto gc) __attribute__((always_inline)) {
    constexpr int g = decltype(gc)::value, kb = g / 3, grp = g % 3;
    constexpr int n_terms = grp + 1, first = grp * (grp + 1) / 2;
    if constexpr (g + 1 < G) {
      constexpr int kn = (g + 1) / 3, gn = (g + 1) % 3;
      static_for<NT>([&](auto itc) __attribute__((always_inline)) {
        pa[(g + 1) & 1][decltype(itc)::value] = lda(std::integral_constant<int, 2 - gn>{}, itc, std::integral_constant<int, kn>{});
      });
      __builtin_amdgcn_sched_barrier(0);
    } else if constexpr (TAIL) {
      load_tail(std::integral_constant<int, 0>{});
      __builtin_amdgcn_sched_barrier(0);
    }
    const Split8p& b = bs(std::integral_constant<int, kb>{});
    static_for<n_terms * NT>([&](auto oc) __attribute__((always_inline)) {
      constexpr int o = decltype(oc)::value, tg = o / NT, it = o % NT;
      constexpr int term = first + tg;  // 0: Al dh | 1: Am dm, 2: Am dh | 3: Ah dl, 4: Ah dm, 5: Ah dh
      const bf16x8 db = __builtin_bit_cast(bf16x8, (term == 0 || term == 2 || term == 5) ? b.h : ((term == 1 || term == 4) ? b.m : b.l));
      if constexpr (SETS == 2 && (term & 1)) extra = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[g & 1][it], db, extra, 0, 0, 0);
      else out[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[g & 1][it], db, out[it], 0, 0, 0);
      fill(std::integral_constant<int, (kb * 6 + term) * NT + it>{});
      __builtin_amdgcn_sched_barrier(0);
    });
  });
  if constexpr (TAIL) {
    static_for<NT>([&](auto itc) __attribute__((always_inline)) {
      constexpr int it = decltype(itc)::value;
      if constexpr (it + 1 < NT) {
        load_tail(std::integral_constant<int, it + 1>{});
        __builtin_amdgcn_sched_barrier(0);
      }
      static_for<12>([&](auto qc) __attribute__((always_inline)) {
        constexpr int q = decltype(qc)::value, kbl = q / 6, term = q % 6, grp = term == 0 ? 0 : (term < 3 ? 1 : 2);
        const Split8p& b = bs(std::integral_constant<int, KBH + kbl>{});
        const bf16x8 db = __builtin_bit_cast(bf16x8, (term == 0 || term == 2 || term == 5) ? b.h : ((term == 1 || term == 4) ? b.m : b.l));
        out[it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pt[it & 1][3 * kbl + grp], db, out[it], 0, 0, 0);
        fill(std::integral_constant<int, KBH * 6 * NT + it * 12 + q>{});
        __builtin_amdgcn_sched_barrier(0);
      });
    });
  }
  if constexpr (SETS == 2) out[0] += extra;
}

// the ordinals of contract_pipe<NT, KB, TAIL> at which tile `it` of the tail is being accumulated: [tail_begin + 12 it, + 12)
template <int NT, int KB>
constexpr int tail_begin() { return (KB - 2) * 6 * NT; }

// Slots of a tile epilogue behind the MFMAs of one block (two K-blocks = NMB MFMAs): MFMA ob of the block runs slots
// [48 ob / NMB, 48 (ob + 1) / NMB) of the 48 the epilogue of a 16-register tile is cut into.
template <int NMB, int OB, class Slot>
__device__ __forceinline__ void run_slots(Slot slot) {
  constexpr int s0 = 48 * OB / NMB, s1 = 48 * (OB + 1) / NMB;
  static_for<s1 - s0>([&](auto k) __attribute__((always_inline)) { slot(std::integral_constant<int, s0 + decltype(k)::value>{}); });
}

// The A operands of the two walks over an image of width C with R rows (split stride R 2 C).
template <int R, int C>
struct Walk {
  static constexpr uint32_t SPLIT = (uint32_t)R * 2u * C;
  lds_bytes img;
  uint32_t fwd_lane;   // forward: row m, unit h
  uint32_t bwd_lane;   // transposed: row 4 h + (i >> 2), columns 16 g + 4 (i & 3) .. + 3 of output tile 0, K-block 0, s = 0
  uint32_t bwd_swz;    // the row's swizzle (independent of kb and it)
  __device__ __forceinline__ Walk(lds_bytes base, int lane) : img(base) {
    const uint32_t m = lane & 31, h = lane >> 5, i = lane & 15, g = (lane >> 4) & 1;
    fwd_lane = m * (uint32_t)Img<C>::RB;
    fwd_swz_ = (16u * h) ^ Img<C>::swz(m);
    const uint32_t row = 4u * h + (i >> 2);
    bwd_lane = row * (uint32_t)Img<C>::RB;
    bwd_swz = Img<C>::swz(row);  // rows 16 kb + 8 s + this: the swizzle of s is folded in below
    bwd_col_ = 32u * g + 16u * (i & 1u) + 8u * ((i >> 1) & 1u);
  }
  uint32_t fwd_swz_, bwd_col_;
  // The images never change after staging, so every operand load is invariant over the step loop: left visible, the
  // compiler hoists them all (a whole image per wave) and spills.  An opaque copy of the lane offsets per evaluation keeps
  // the loads where their MFMAs are.
  __device__ __forceinline__ Walk opaque() const {
    Walk w = *this;
    asm volatile("" : "+v"(w.fwd_lane), "+v"(w.bwd_lane));
    return w;
  }
  // forward: row 32 it + m at the columns of (kb, h)
  template <int SP, int IT, int KBI>
  __device__ __forceinline__ bf16x8 fwd(std::integral_constant<int, SP>, std::integral_constant<int, IT>, std::integral_constant<int, KBI>) const {
    const uint32_t off = fwd_lane + (fwd_swz_ ^ (32u * KBI));
    return *(lds_bf16x8)(img + (uint32_t)SP * SPLIT + (uint32_t)IT * 32u * Img<C>::RB + off);
  }
  // transposed: column 32 it + m at the rows of (kb, h)
  template <int SP, int IT, int KBI>
  __device__ __forceinline__ bf16x8 bwd(std::integral_constant<int, SP>, std::integral_constant<int, IT>, std::integral_constant<int, KBI>) const {
    constexpr uint32_t rows0 = 16u * KBI;
    // swizzle of row 16 kb + 8 s + (4 h + A): the row bits above bit 3 never enter it; bit 3 (s) does for C = 128 / 64 / 32
    // through v = (row >> LB): fold the compile-time part
    constexpr uint32_t s1 = Img<C>::swz(8u);  // the swizzle contribution of s = 1 (XOR-linear in the row bits)
    const uint32_t c0 = (64u * IT) ^ 0u, c1 = (64u * IT) ^ s1;
    const uint32_t lane_x = bwd_col_ ^ bwd_swz;
    const bf16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (lds_bf16x4)(img + (uint32_t)SP * SPLIT + rows0 * Img<C>::RB + bwd_lane + (lane_x ^ c0)));
    const bf16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(
        (lds_bf16x4)(img + (uint32_t)SP * SPLIT + (rows0 + 8u) * Img<C>::RB + bwd_lane + (lane_x ^ c1)));
    return __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
  }
};

}  // namespace mlpb16
}  // namespace ebm
