// The dense Gaussian contraction  g = Ps (x - mu)  on the bf16 matrix pipe at fp32 accuracy.
//
// The exact-f32 MFMA (v_mfma_f32_32x32x2_f32) runs at the f32 VECTOR rate and on the same lanes as the VALU: it never
// overlaps the Philox / Box-Muller work of a Langevin step, and a dim-128 evaluation is 256 instructions of 64 cycles.
// v_mfma_f32_32x32x16_bf16 is 16x that rate and a separate pipe.  Both operands are split into three bf16 pieces
//   v = hi + mid + lo,   hi = bf16(v), mid = bf16(v - hi), lo = bf16(v - hi - mid)      (3 x 8 = 24 mantissa bits)
// and the six products of total order <= 2 are accumulated in fp32 by the matrix pipe:
//   d.P ~= dh.Ph + (dh.Pm + dm.Ph) + (dh.Pl + dm.Pm + dl.Ph);   the dropped terms are below 2^-24 relative.
// A bf16 x bf16 product is exact in fp32, so what is left is fp32 accumulation error -- the same class as the f32 MFMA
// (tests: Gaussian tolerance of tests/test_langevin_gpu.py, unchanged).  6/16 of the f32 matrix time, off the VALU.
//
// Layouts.  The chain state lives in the C/D layout of 32x32 tiles (gauss_mfma.hip): lane (m, h), register r of tile t
// = coordinate 32 t + (r & 3) + 8 (r >> 2) + 4 h of chain m.  A K-block of the bf16 instruction is 16 coordinates; lane
// half h supplies 8 of them, element j of its operand pairing with element j of the other operand's same half -- which
// eight coordinates those are is ours to choose as long as A and B agree.  We take registers 8 b .. 8 b + 7 of tile t
// (K-block kb = 2 t + b): eight CONSECUTIVE state registers, no shuffling.  The matching A operand -- row 32 it + m of
// Ps at those eight columns -- is laid out in LDS operand-ready: Aop[split][it][kb][lane][8] bf16, one ds_read_b128
// per lane, conflict-free.
#pragma once
#include <type_traits>
#include <utility>

#include "ebm_common.h"

namespace ebm {
namespace gauss3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): a loop whose index is a compile-time constant in
// the body (register arrays indexed by it stay registers; `if constexpr` on it works)
template <class F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, I>{}), ...);
}
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(static_cast<F&&>(f), std::make_integer_sequence<int, N>{});
}

// bytes of LDS for the three splits of a (32 NT)^2 matrix
__host__ __device__ constexpr size_t aop_bytes(int NT) { return (size_t)3 * (32 * NT) * (32 * NT) * 2; }

// coordinate (inside the padded DIM) that element j of lane-half h holds in K-block kb
__device__ __forceinline__ int k_of(int kb, int h, int j) { return 32 * (kb >> 1) + 16 * (kb & 1) + 8 * (j >> 2) + 4 * h + (j & 3); }

__device__ __forceinline__ void split3(float v, __bf16& hi, __bf16& mid, __bf16& lo) {
  hi = (__bf16)v;
  const float r1 = v - (float)hi;
  mid = (__bf16)r1;
  lo = (__bf16)(r1 - (float)mid);
}

// All threads of the workgroup: an [32 MT] x [16 KB] matrix given by `at(row, col)` -> its three operand-ready splits in
// LDS, Aop[split][it][kb][lane][8]: lane (m, h) of (it, kb) holds row 32 it + m at the columns k_of(kb, h, 0..7).
template <int MT, int KB, class At>
__device__ __forceinline__ void stage_split_matrix(At&& at, __bf16* aop, int n_threads) {
  constexpr int PER_SPLIT = MT * KB * 64 * 8;
  for (int i = threadIdx.x; i < PER_SPLIT; i += n_threads) {
    const int j = i & 7, lane = (i >> 3) & 63, kb = (i >> 9) % KB, it = (i >> 9) / KB;
    const float v = at(32 * it + (lane & 31), k_of(kb, lane >> 5, j));
    __bf16 hi, mid, lo;
    split3(v, hi, mid, lo);
    aop[i] = hi;
    aop[PER_SPLIT + i] = mid;
    aop[2 * PER_SPLIT + i] = lo;
  }
}
__host__ __device__ constexpr size_t aop_bytes_general(int MT, int KB) { return (size_t)3 * MT * KB * 64 * 8 * 2; }

// prec[dim][dim] (symmetric, fp32, global), zero-padded to (32 NT)^2
template <int NT>
__device__ __forceinline__ void stage_split_precision(const float* __restrict__ prec, int dim, __bf16* aop, int n_threads) {
  stage_split_matrix<NT, 2 * NT>([&](int row, int col) { return (row < dim && col < dim) ? prec[row * dim + col] : 0.0f; }, aop,
                                 n_threads);
}

// out (C/D layout, MT tiles, overwritten) = A (b - mus) for the [32 MT] x [16 KB] matrix A staged by stage_split_matrix;
// b: the K operand in the C/D layout of its own 32-row tiles (K-block kb = registers 8 (kb & 1) .. + 7 of tile kb >> 1);
// SUB: subtract mus[k] (fp32 [16 KB] in LDS) first.
// `fill()` is called once behind every MFMA: the caller's independent VALU work (the Philox rounds of the step), fenced
// so that it stays there -- it issues while the matrix pipe is busy with that MFMA (32 cycles each).
struct NoFill {
  template <class Ord>
  __device__ __forceinline__ void operator()(Ord) const {}
};
// MTL / IT0: the staged matrix has MTL row tiles and this call produces tiles IT0 .. IT0 + MT - 1 of them (MT < MTL:
// the output in pieces, for bodies that have no registers for all accumulators and operands at once).
// KBU: the K-blocks actually contracted (the staged layout keeps KB): trailing K-blocks that hold nothing but zero padding
// -- 16 KBU >= dim -- add exactly 0 to every accumulator and are left out (their MFMAs, their split of the operand).
template <int MT, int KB, bool SUB, class Fill = NoFill, int MTL = MT, int IT0 = 0, int KBU = KB>
__device__ __forceinline__ void contract_general(const __bf16* __restrict__ aop, const float* __restrict__ mus,
                                                 const f32x16 (&x)[(KB + 1) / 2], f32x16 (&g)[MT], int lane,
                                                 Fill&& fill = NoFill{}) {
  constexpr int NT = MT, PER_SPLIT = MTL * KB * 64 * 8;
  const int h = lane >> 5;
  // one tile: two accumulator sets, so that consecutive MFMAs never wait on each other's result; with more tiles the
  // term-major order below already puts NT independent instructions between dependent ones
  constexpr int SETS = NT == 1 ? 2 : 1;
  f32x16 acc[SETS][NT];
#pragma unroll
  for (int s = 0; s < SETS; ++s)
#pragma unroll
    for (int it = 0; it < NT; ++it)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[s][it][r] = 0.0f;
  const bf16x8* ap = reinterpret_cast<const bf16x8*>(aop) + lane;
  static_assert(KBU >= 1 && KBU <= KB, "K-blocks used");
  static_for<KBU>([&](auto kbc) {
    constexpr int kb = decltype(kbc)::value;
    constexpr int t = kb >> 1, b = kb & 1;
    float4 m0 = make_float4(0.f, 0.f, 0.f, 0.f), m1 = m0;
    if constexpr (SUB) {
      m0 = *reinterpret_cast<const float4*>(mus + 32 * t + 16 * b + 4 * h);
      m1 = *reinterpret_cast<const float4*>(mus + 32 * t + 16 * b + 8 + 4 * h);
    }
    f32x8 d;
    d[0] = x[t][8 * b + 0] - m0.x; d[1] = x[t][8 * b + 1] - m0.y; d[2] = x[t][8 * b + 2] - m0.z; d[3] = x[t][8 * b + 3] - m0.w;
    d[4] = x[t][8 * b + 4] - m1.x; d[5] = x[t][8 * b + 5] - m1.y; d[6] = x[t][8 * b + 6] - m1.z; d[7] = x[t][8 * b + 7] - m1.w;
    const bf16x8 dh = __builtin_convertvector(d, bf16x8);
    const f32x8 r1 = d - __builtin_convertvector(dh, f32x8);
    const bf16x8 dm = __builtin_convertvector(r1, bf16x8);
    const f32x8 r2 = r1 - __builtin_convertvector(dm, f32x8);
    const bf16x8 dl = __builtin_convertvector(r2, bf16x8);
    // six terms, smallest first, grouped by the split of Ps they read (one split's operands live at a time: lo for one
    // term, mid for two, hi for three); inside a group term-major, so NT independent MFMAs sit between dependent ones
    constexpr int S1 = SETS - 1;
    static_for<3>([&](auto gc) {
      constexpr int grp = decltype(gc)::value;            // 0: Pl, 1: Pm, 2: Ph
      constexpr int n_terms = grp + 1, first = grp * (grp + 1) / 2;
      bf16x8 pa[NT];
#pragma unroll
      for (int it = 0; it < NT; ++it) pa[it] = ap[(2 - grp) * (PER_SPLIT / 8) + ((IT0 + it) * KB + kb) * 64];
      static_for<n_terms * NT>([&](auto oc) {
        constexpr int o = decltype(oc)::value, tg = o / NT, it = o % NT;
        constexpr int term = first + tg;                  // 0: Pl dh | 1: Pm dm, 2: Pm dh | 3: Ph dl, 4: Ph dm, 5: Ph dh
        constexpr int set = (term & 1) ? S1 : 0;
        const bf16x8& db = (term == 0 || term == 2 || term == 5) ? dh : ((term == 1 || term == 4) ? dm : dl);
        acc[set][it] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[it], db, acc[set][it], 0, 0, 0);
        if constexpr (!std::is_same<std::decay_t<Fill>, NoFill>::value) __builtin_amdgcn_sched_barrier(0);  // (round 6: the MFMA first, then its fill)
        fill(std::integral_constant<int, (kb * 6 + term) * NT + it>{});
      });
    });
  });
#pragma unroll
  for (int it = 0; it < NT; ++it) {
    if constexpr (SETS == 2) g[it] = acc[0][it] + acc[1][it];
    else g[it] = acc[0][it];
  }
}

// The Gaussian form with the OUTPUT in pieces of P tiles: all accumulators, the A operands of every tile and the split's
// transients do not fit beside an HMC body's resident state (x, p, force) at once; the split of x is formed once per piece
// instead, every time after the first from an OPAQUE copy of x -- left visible, the compiler merges the identical splits and
// keeps the 12 split registers of every K-block live across the pieces (1 - 2 KB of scratch at three / four tiles).
// `fill` ordinals run over all pieces: 0 .. 6 NT KBU - 1, as in the one-piece form.
template <int NT, int P, int KBU = 2 * NT, class Fill = NoFill>
__device__ __forceinline__ void contract_pieces(const __bf16* __restrict__ aop, const float* __restrict__ mus, const f32x16 (&x)[NT],
                                                f32x16 (&g)[NT], int lane, Fill&& fill = NoFill{}) {
  static_assert(P >= 1 && P < NT, "more than one piece");
  constexpr int KB = 2 * NT, PIECES = (NT + P - 1) / P;
  static_for<PIECES>([&](auto pc) {
    constexpr int pi = decltype(pc)::value, it0 = pi * P, mt = (NT - it0) < P ? (NT - it0) : P;
    f32x16 xb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      xb[t] = x[t];
      if constexpr (pi > 0) asm volatile("" : "+v"(xb[t]));
    }
    f32x16 out[mt];
    auto shifted = [&](auto ord) { fill(std::integral_constant<int, 6 * it0 * KBU + decltype(ord)::value>{}); };
    contract_general<mt, KB, true, decltype(shifted)&, NT, it0, KBU>(aop, mus, xb, out, lane, shifted);
#pragma unroll
    for (int t = 0; t < mt; ++t) g[it0 + t] = out[t];
  });
}

// the Gaussian form: g = Ps (x - mu) on (32 NT)^2
template <int NT, int KBU = 2 * NT, class Fill = NoFill>
__device__ __forceinline__ void contract(const __bf16* __restrict__ aop, const float* __restrict__ mus, const f32x16 (&x)[NT],
                                         f32x16 (&g)[NT], int lane, Fill&& fill = NoFill{}) {
  contract_general<NT, 2 * NT, true, Fill, NT, 0, KBU>(aop, mus, x, g, lane, static_cast<Fill&&>(fill));
}

}  // namespace gauss3
}  // namespace ebm
