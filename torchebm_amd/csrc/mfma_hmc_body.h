// (shared by gauss_hmc_mfma.hip, gmm_hmc_mfma.hip and matrix_hmc_diag.hip)
// HMC transitions for the dense Gaussian energy on the matrix cores (dim 32 or 64, no mass or a scalar
// mass): same mapping as gauss_mfma.hip -- a wavefront owns 32 chains, lane l = (m, h), the chain
// state x, momentum p and force f all live in the C/D layout of v_mfma_f32_32x32x2_f32 tiles, so that
// g^T = Ps (x - mu)^T takes its B-operand straight from the state registers and lands in the layout the
// leapfrog arithmetic runs in.  The LDS mat-vec of the lane-group kernel (rows.h) is bound by LDS reads
// (one 16-byte read per four FMAs); here the precision matrix is read once per K-step for 32 chains.
//
// Reference: samplers/hmc.py:201-315 (transition), integrators/leapfrog.py:116-187 (safe leapfrog),
// core/base_model.py:181-210 (energy).  Semantics, RNG coordinates (momentum at step 2t, uniforms at
// 2t+1, one Philox counter per four coordinates) and the fast/literal split of the safe mode are those
// of hmc_kernel.h; the accepted state is "parked" in the x array itself (written on accept, re-read on
// reject), which costs 256 B per chain and transition and no registers.
#pragma once
#include <type_traits>
#include "diag.h"
#include "ebm_common.h"
#include "gauss_bf16x3.h"
#include "gmm_bf16x3.h"

namespace ebm {
namespace {

constexpr int kBlock = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GaussHmcArgs {
  float* x;
  int64_t n_chains;
  int32_t dim;
  int32_t n_mh, n_leapfrog;
  float eps;
  const float* eps_table;
  int32_t has_mass;
  float mass_raw, mass_sqrt, mass_safe;
  int32_t thin, n_kept;
  float* traj;
  uint8_t* accept_mask;
  uint32_t* accept_count;
  const float* p_noise;
  const float* u;
  RngKey key;
  uint64_t step0;
  const float* mean;  // [dim]
  const float* prec;  // [dim, dim], symmetric
  const float* mass_diag;  // [dim] diagonal mass (null: none / scalar)
  const char* prec_image;  // dense Gaussian beyond 160 dims (gauss_hmc_stream.hip): the pre-split precision image, else null
  // Gaussian mixture (GmmE below): means [n_comp, dim], log-weights [n_comp], 1 / (2 sigma^2), 1 / sigma^2
  const float* gm_means;
  const float* gm_logw;
  int32_t n_comp;
  float inv2s2, invs2;
  diag::DiagArgs diag;     // per-workgroup diagnostics records at the kept transitions (DIAG kernels)
  int32_t sh_classes = 1;  // SHIFTED rows (SH kernels): 4 / gcd(dim, 4) alignment classes of chains, one per workgroup
  int32_t sh_lo = 0;       // ... and, set by the body in its own copy, this workgroup's row offset (what the energies see)
  int64_t sh_image_stride = 0;  // bytes between the classes' pre-split images (the streamed evaluation on shifted rows)
  int32_t tr0 = 0;         // first transition to run (PW kernels hand a workgroup over to the literal body in mid-call: see gauss_hmc_fallback)
  int32_t resumed = 0;     // ... a single WAVE does (energies without barriers inside): LDS is staged, no workgroup barrier may be met
};

extern __shared__ __attribute__((aligned(16))) float gauss_hmc_smem[];

template <int NT>
struct Tile {
  f32x16 t[NT];
};

// g^T = Ps (x - mu)^T and E = 0.5 (x - mu)^T Ps (x - mu) per chain (both halves of the wave hold E).
// B3: the contraction on the bf16 matrix pipe with three-way split operands (gauss_bf16x3.h; `Ps` then points at the
// operand-ready splits) -- 6/16 of the exact-f32 MFMA's matrix time; B3 = false: v_mfma_f32_32x32x2_f32 on fp32 Ps.
// KT: trailing all-padding K-blocks left out of the bf16 contraction (gauss_bf16x3.h KBU)
template <int NT, bool B3, int KT = 0>
__device__ __forceinline__ float gauss_eval(const float* Ps, const float* mus, const Tile<NT>& x, Tile<NT>& g, int m, int h) {
  constexpr int DIM = 32 * NT;
  auto k_of = [&](int s) { return 32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2) + 4 * h; };
  float acc = 0.0f;
  if constexpr (B3 && NT >= 3) {
    // the output in two pieces of at most two tiles: gauss_bf16x3.h says why.  (Two tiles in single-tile pieces fit 256
    // VGPRs / two waves per SIMD without spills, but run 0.54 - 0.58 ms per 10 transitions at dims 48 / 64 where the one-piece
    // form, unconstrained, runs 0.46 - 0.50.)
    gauss3::contract_pieces<NT, 2, 2 * NT - KT>(reinterpret_cast<const __bf16*>(Ps), mus, x.t, g.t, m + 32 * h);
  } else if constexpr (B3) {
    gauss3::contract<NT, 2 * NT - KT>(reinterpret_cast<const __bf16*>(Ps), mus, x.t, g.t, m + 32 * h);
  } else {
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) g.t[t][r] = 0.0f;
  float pa[NT], pb[NT], ma, mb;
#pragma unroll
  for (int it = 0; it < NT; ++it) pa[it] = Ps[k_of(0) * DIM + 32 * it + m];
  ma = mus[k_of(0)];
#pragma unroll
  for (int s = 0; s < 16 * NT; ++s) {  // operands of K-step s+1 are requested before the MFMAs of K-step s issue
    if (s + 1 < 16 * NT) {
      const int kn = k_of(s + 1);
#pragma unroll
      for (int it = 0; it < NT; ++it) pb[it] = Ps[kn * DIM + 32 * it + m];
      mb = mus[kn];
    }
    const float dv = x.t[s >> 4][s & 15] - ma;
#pragma unroll
    for (int it = 0; it < NT; ++it) g.t[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[it], dv, g.t[it], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int it = 0; it < NT; ++it) pa[it] = pb[it];
    ma = mb;
  }
  }  // exact-f32 MFMA
  // E = 0.5 d.g with d = x - mu recomputed (a cheap LDS read of mu, two distinct addresses per wave) rather
  // than kept: 16*NT fewer live registers across the MFMA loop
#pragma unroll
  for (int s = 0; s < 16 * NT; ++s) acc = __builtin_fmaf(x.t[s >> 4][s & 15] - mus[k_of(s)], g.t[s >> 4][s & 15], acc);
  acc += __shfl_xor(acc, 32);
  return 0.5f * acc;
}

// ---------------------------------------------------------------------------------
// Energies of the matrix-layout transition body: what sits in LDS and how E / dE/dx come out of the state tiles.
// ---------------------------------------------------------------------------------
// Dense Gaussian.  LDS: Ps (fp32 [DIM][DIM], or its three operand-ready bf16 splits), mu [DIM].
template <int NT, bool B3, int KT = 0>
struct GaussE {
  static_assert(KT == 0 || B3, "trimmed K-blocks: the bf16 contraction");
  static constexpr int DIM = 32 * NT;
  static constexpr int kMatFloats = B3 ? (int)(gauss3::aop_bytes(NT) / sizeof(float)) : DIM * DIM;
  static constexpr int kLdsFloats = kMatFloats + DIM;
  static constexpr bool kEvalGivesEnergy = true;
  static constexpr bool kCarry = !(B3 && NT >= 4);  // (four / five tiles: the split operands leave no LDS for the parked force)
  // lo: SHIFTED rows -- the row's column 0 sits at tile coordinate lo (gauss_mfma_body.h SH)
  __device__ static __forceinline__ void stage(const GaussHmcArgs& a, float* lds, int lo) {
    static_assert(B3, "shifted rows: the bf16 contraction");
    const int dim = a.dim;
    gauss3::stage_split_matrix<NT, 2 * NT>([&](int r, int c) {
      r -= lo; c -= lo;
      return (r >= 0 && c >= 0 && r < dim && c < dim) ? a.prec[r * dim + c] : 0.0f;
    }, reinterpret_cast<__bf16*>(lds), kBlock);
    for (int i = threadIdx.x; i < DIM; i += kBlock) lds[kMatFloats + i] = (i >= lo && i - lo < dim) ? a.mean[i - lo] : 0.0f;
  }
  __device__ static __forceinline__ void stage(const GaussHmcArgs& a, float* lds) {
    const int dim = a.dim;
    if constexpr (B3) {
      gauss3::stage_split_precision<NT>(a.prec, dim, reinterpret_cast<__bf16*>(lds), kBlock);
    } else {
      for (int i = threadIdx.x; i < DIM * DIM; i += kBlock) {
        const int r = i / DIM, c = i - r * DIM;
        lds[i] = (r < dim && c < dim) ? a.prec[r * dim + c] : 0.0f;
      }
    }
    for (int i = threadIdx.x; i < DIM; i += kBlock) lds[kMatFloats + i] = i < dim ? a.mean[i] : 0.0f;
  }
  __device__ static __forceinline__ float eval(const GaussHmcArgs&, const float* lds, const Tile<NT>& x, Tile<NT>& g, int m, int h) {
    return gauss_eval<NT, B3, KT>(lds, lds + kMatFloats, x, g, m, h);
  }
  __device__ static __forceinline__ float energy(const GaussHmcArgs&, const float*, const Tile<NT>&, int, int) { return 0.0f; }
  // (round 6) the force in PIECES for the transition body's PW path (four / five tiles on the split contraction: position,
  // momentum and force are 192 / 240 registers beside ~150 of operands): eval() already forms the output two tiles at a time
  // (contract_pieces) -- here a piece is handed out as soon as it is done, and the body kicks its momentum tiles at once.
  static constexpr bool kPiecewise = B3 && NT >= 4;
  static constexpr int kPieces = (NT + 1) / 2;
  static constexpr int piece_t0(int pi) { return 2 * pi; }
  static constexpr int piece_tn(int pi) { return NT - 2 * pi < 2 ? NT - 2 * pi : 2; }
  template <int T0, int TN>
  __device__ __forceinline__ float eval_tiles(const GaussHmcArgs&, const float* lds, const Tile<NT>& x, f32x16 (&gout)[TN], int m, int h,
                                              bool want_e = true) const {
    const float* mus = lds + kMatFloats;
    f32x16 xb[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      xb[t] = x.t[t];
      if constexpr (T0 > 0) asm volatile("" : "+v"(xb[t]));  // (an opaque copy: gauss_bf16x3.h, contract_pieces)
    }
    gauss3::contract_general<TN, 2 * NT, true, gauss3::NoFill, NT, T0, 2 * NT - KT>(reinterpret_cast<const __bf16*>(lds), mus, xb, gout, m + 32 * h);
    if (!want_e) return 0.0f;
    float acc = 0.0f;
#pragma unroll
    for (int s = 16 * T0; s < 16 * (T0 + TN); ++s) {
      const int k = 32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2) + 4 * h;
      acc = __builtin_fmaf(x.t[s >> 4][s & 15] - mus[k], gout[(s >> 4) - T0][s & 15], acc);
    }
    acc += __shfl_xor(acc, 32);
    return 0.5f * acc;
  }
};

// Isotropic Gaussian mixture, up to 32 components: gmm_bf16x3.h (both K x dim passes of the gradient on the bf16 matrix
// pipe; the energy -- needed twice per transition -- in the reference's difference form on the VALU).
template <int NT, int KR>
struct GmmE {
  using M = gmm3::Mixture<NT, KR>;
  static constexpr int kLdsFloats = M::kLdsFloats;
  static constexpr bool kEvalGivesEnergy = false;
  static constexpr bool kCarry = NT <= 4;  // (five to eight tiles: the parked force, 16 KB per tile, does not fit beside the operands)
  __device__ static __forceinline__ gmm3::Params params(const GaussHmcArgs& a) {
    return gmm3::Params{a.gm_means, a.gm_logw, a.n_comp, a.dim, a.inv2s2, a.invs2, a.sh_lo};
  }
  __device__ static __forceinline__ void stage(const GaussHmcArgs& a, float* lds) { M::stage(params(a), lds, kBlock); }
  __device__ static __forceinline__ void stage(const GaussHmcArgs& a, float* lds, int lo) {  // shifted rows
    gmm3::Params pr = params(a);
    pr.lo = lo;
    M::stage(pr, lds, kBlock);
  }
  // gradient into g; returns the softmax sum (in [1, K] for a finite state, NaN as soon as a coordinate is not)
  __device__ static __forceinline__ float eval(const GaussHmcArgs& a, const float* lds, const Tile<NT>& x, Tile<NT>& g, int m, int h) {
    return M::grad(params(a), lds, x.t, g.t, m + 32 * h);
  }
  __device__ static __forceinline__ float energy(const GaussHmcArgs& a, const float* lds, const Tile<NT>& x, int m, int h) {
    return M::energy(params(a), lds, x.t, m + 32 * h);
  }
  // (round 6) the force in PIECES of two output tiles for the transition body's PW path (five tiles and more: nothing is carried):
  // the softmax weights once per evaluation (kept here between the pieces), the weighted mean and the gradient piece by piece
  // (up to 16 components: measured -3 ... -20 % at dims 129 ... 224, 2^16 chains x 4 transitions x L = 10; with 17 ... 32 -- two
  //  K-blocks of weights -- the pieces cost +9 ... +42 % and the one-piece form stays)
  static constexpr bool kPiecewise = !kCarry && (KR <= 8 || NT == 8);  // (eight tiles: in pieces or not at all)
  static constexpr int kPieces = (NT + 1) / 2;
  static constexpr int piece_t0(int pi) { return 2 * pi; }
  static constexpr int piece_tn(int pi) { return NT - 2 * pi < 2 ? NT - 2 * pi : 2; }
  f32x16 w_[1];
  float sum_ = 0.0f;
  template <int T0, int TN>
  __device__ __forceinline__ float eval_tiles(const GaussHmcArgs& a, const float* lds, const Tile<NT>& x, f32x16 (&gout)[TN], int m, int h,
                                              bool want_e = true) {
    if constexpr (T0 == 0) sum_ = M::weights(params(a), lds, x.t, w_, m + 32 * h);
    M::template grad_tiles<T0, TN>(params(a), lds, x.t, w_, sum_, gout, m + 32 * h);
    if constexpr (T0 == 0) return want_e ? M::energy(params(a), lds, x.t, m + 32 * h) : 0.0f;  // (the exact energy, once per evaluation that wants it)
    else return 0.0f;
  }
};

// DIAGM: diagonal mass (its own instantiation: as a run-time switch it cost the plain kernels their register allocation)
// E: the energy (GaussE / GmmE above).
// DIAG: emit the in-kernel diagnostics records (diag.h: wave_record, one per wave of 32 chains, straight from the C/D
// registers) at the kept transitions; the energy is the carried one, the accept share the decision just taken.
// CARRY off (E::kCarry: the four-tile Gaussian on the split contraction, whose 96 KB of operands leave no LDS for the parked
// force): energy and force are evaluated at the top of every transition, as the reference does.
// An energy whose evaluation has workgroup barriers inside (GaussStreamE: the slabs of Ps stream through LDS) says so with
// kBlockVote: the fast / literal decision of a leapfrog step is then taken per WORKGROUP -- every wave makes the same number
// of evaluations.  (For chains that are fine the literal path computes exactly what the fast path does.)
template <class E, class = void>
struct BlockVote { static constexpr bool value = false; };
template <class E>
struct BlockVote<E, std::enable_if_t<E::kBlockVote>> { static constexpr bool value = true; };
template <class E>
__device__ __forceinline__ bool vote_all(bool pred) {
  if constexpr (BlockVote<E>::value) return __syncthreads_and(pred ? 1 : 0) != 0;
  else return __all(pred);
}

// SH: SHIFTED rows (widths off multiples of 4; gauss_mfma_body.h says how): a workgroup takes the chains of one alignment
// class, tile coordinate j = coordinate j - lo of the chain; the tile coordinates outside [lo, lo + dim) are padding like
// the ones beyond dim -- x = p = f = 0 throughout (loaded as 0, their momentum draw discarded, zero rows of the staged matrix).
// (round 6) PW -- the force in PIECES (energies that offer eval_tiles<T0, TN>, kPieces and piece_t0 / piece_tn: GaussStreamE, GaussE from four tiles).  At seven /
// eight tiles position + momentum + force are 336 / 384 of a wave's 512 registers (of which only 256 can be operands of vector
// instructions) and the allocator kept ~330 values per lane in scratch for the whole trajectory, moving them through every kick and
// drift: 41 GB of traffic per launch at dim 256 (profiles/r05_pmc.json), 97 k cycles per evaluation against 24.6 k of MFMAs.  With PW a
// leapfrog step evaluates the force one piece of output tiles at a time (a pass over the streamed matrix each) and kicks the momentum
// of that piece at once -- p += eps f between interior steps, eps / 2 at the trajectory's ends: the two half kicks of the reference
// merged, as in the element-wise fast body (hmc_kernel.h) -- so no force array outlives its piece.  Safe mode: the fast sequence is
// what the literal one computes while every energy and momentum stays finite; a workgroup that sees anything else hands the REST of its
// call (from the transition at hand, whose accepted state is in a.x) to the literal body, out of line (gauss_hmc_fallback).
template <class E, class = void>
struct piecewise_of { static constexpr bool value = false; };
template <class E>
struct piecewise_of<E, std::void_t<decltype(E::kPieces)>> { static constexpr bool value = true; };
template <int NT, bool B3, int KT>
struct piecewise_of<GaussE<NT, B3, KT>, void> { static constexpr bool value = GaussE<NT, B3, KT>::kPiecewise; };
template <int NT, int KR>
struct piecewise_of<GmmE<NT, KR>, void> { static constexpr bool value = GmmE<NT, KR>::kPiecewise; };
template <int NT, bool DIAGM, class E, bool DIAG, bool SH>
__device__ __noinline__ void gauss_hmc_fallback(const GaussHmcArgs& a);

template <int NT, bool DIAGM, class E, bool DIAG = false, bool SH = false, bool PW = false>
__device__ __forceinline__ void gauss_hmc_mfma_body(const GaussHmcArgs& a_in) {
  constexpr bool CARRY = E::kCarry;
  const int sh_s = SH ? (int)(blockIdx.x % (unsigned)a_in.sh_classes) : 0;
  const int lo = SH ? ((a_in.dim * sh_s) & 3) : 0;
  GaussHmcArgs a_sh = a_in;  // (SH: the energies see this workgroup's row offset)
  a_sh.sh_lo = lo;
  if (SH && a_in.prec_image) a_sh.prec_image = a_in.prec_image + (int64_t)sh_s * a_in.sh_image_stride;  // this class's image
  const GaussHmcArgs& a = SH ? a_sh : a_in;
  E en{};  // (state of the evaluation across calls, if it has any: GaussStreamE's buffer parity)
  constexpr int DIM = 32 * NT;
  float* elds = gauss_hmc_smem;  // the energy's own area
  // dim <= DIM, dim % 4 == 0: zero-padded tiles -- padded coordinates have x = p = f = 0 throughout (their
  // rows / columns of the parameters are zero, their momentum draw is discarded) and are never loaded or stored
  const int dim = a.dim;
  const int hi = lo + dim;
  // (a single wave that resumes its call in the literal body -- gauss_hmc_fallback, energies without barriers inside -- finds LDS
  //  staged and must not meet a workgroup barrier)
  const bool staged = !BlockVote<E>::value && a.resumed != 0;
  if (!staged) {
    if constexpr (SH) E::stage(a, elds, lo);
    else E::stage(a, elds);
  }
  // Diagonal mass (samplers/hmc.py:136-159, integrators/leapfrog.py:116-149): the raw masses sit in LDS (padded
  // with 1), every lane reads the four of a quad with one broadcast float4; the drift factors eps / max(m, 1e-10)
  // of a transition go through a row of this wave's own (lanes of one K-half hold the same coordinates).
  float* mraw = elds + E::kLdsFloats;                        // [DIM]
  float* dsw_base = mraw + DIM;
  float* dsw = dsw_base + (threadIdx.x >> 6) * DIM;          // [DIM], this wave's
  constexpr bool diag_mass = DIAGM;
  if (!staged) {
    if constexpr (diag_mass)
      for (int i = threadIdx.x; i < DIM; i += kBlock) mraw[i] = (i >= lo && i < hi) ? a.mass_diag[i - lo] : 1.0f;
    __syncthreads();
  }

  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int64_t chain = SH ? (((int64_t)(blockIdx.x / (unsigned)a.sh_classes) * (kBlock / 64) + (threadIdx.x >> 6)) * 32 + m) * a.sh_classes + sh_s
                           : ((int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6)) * 32 + m;
  const bool active = chain < a.n_chains;
  const int64_t row = active ? chain * (int64_t)dim - lo : 0;  // flat element of tile coordinate 0 (SH: a multiple of 4 all the same)
  auto quad_of = [&](const float* arr, int t, int q) { return *reinterpret_cast<const float4*>(arr + 32 * t + 8 * q + 4 * h); };

  // quad q of tile t = coordinates 32t + 8q + 4h .. +3  (one float4, one Philox counter)
  // (SH: `off` addresses tile coordinate 0; quads_aligned: base + off is on the float4 grid -- the state itself; an injected
  //  field or a trajectory row starts anywhere, and a quad shared with a neighbouring chain is element-wise in any case)
  auto load_rows = [&](const float* base, int64_t off, Tile<NT>& dst, bool quads_aligned = true) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int k0 = 32 * t + 8 * q + 4 * h;
        if constexpr (SH) {
          const float* src = base + off + k0;
          if (active && quads_aligned && k0 >= lo && k0 + 3 < hi) {
            v = *reinterpret_cast<const float4*>(src);
          } else if (active && k0 + 3 >= lo && k0 < hi) {
            if (k0 + 0 >= lo && k0 + 0 < hi) v.x = src[0];
            if (k0 + 1 >= lo && k0 + 1 < hi) v.y = src[1];
            if (k0 + 2 >= lo && k0 + 2 < hi) v.z = src[2];
            if (k0 + 3 >= lo && k0 + 3 < hi) v.w = src[3];
          }
        } else
        if (active && 32 * t + 8 * q + 4 * h < dim) v = *reinterpret_cast<const float4*>(base + off + 32 * t + 8 * q + 4 * h);
        dst.t[t][4 * q] = v.x; dst.t[t][4 * q + 1] = v.y; dst.t[t][4 * q + 2] = v.z; dst.t[t][4 * q + 3] = v.w;
      }
  };
  auto store_rows = [&](float* base, int64_t off, const Tile<NT>& src, bool quads_aligned = true) {
    if (!active) return;
    if constexpr (SH) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k0 = 32 * t + 8 * q + 4 * h;
          float* dst = base + off + k0;
          if (quads_aligned && k0 >= lo && k0 + 3 < hi) {
            *reinterpret_cast<float4*>(dst) = make_float4(src.t[t][4 * q], src.t[t][4 * q + 1], src.t[t][4 * q + 2], src.t[t][4 * q + 3]);
          } else if (k0 + 3 >= lo && k0 < hi) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (k0 + i >= lo && k0 + i < hi) dst[i] = src.t[t][4 * q + i];
          }
        }
      return;
    }
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (32 * t + 8 * q + 4 * h < dim)
          *reinterpret_cast<float4*>(base + off + 32 * t + 8 * q + 4 * h) =
              make_float4(src.t[t][4 * q], src.t[t][4 * q + 1], src.t[t][4 * q + 2], src.t[t][4 * q + 3]);
  };
  // K(p) = 0.5 p^T p [/ m], clamped to [0, 1e10]  (samplers/hmc.py:136-159, :251-254)
  auto kinetic = [&](const Tile<NT>& q) -> float {
    float acc = 0.0f;
    if constexpr (diag_mass) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 mq = quad_of(mraw, t, qd);
          acc += q.t[t][4 * qd] * q.t[t][4 * qd] / mq.x;
          acc += q.t[t][4 * qd + 1] * q.t[t][4 * qd + 1] / mq.y;
          acc += q.t[t][4 * qd + 2] * q.t[t][4 * qd + 2] / mq.z;
          acc += q.t[t][4 * qd + 3] * q.t[t][4 * qd + 3] / mq.w;
        }
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc += q.t[t][r] * q.t[t][r];
    }
    acc += __shfl_xor(acc, 32);
    float k = 0.5f * acc;
    if (a.has_mass) k = k / a.mass_raw;
    return clamp_nanprop(k, 0.0f, 1e10f);
  };

  Tile<NT> x;
  load_rows(a.x, row, x);
  const int64_t traj_row = active ? chain * (int64_t)a.n_kept * dim - lo : 0;
  int until_keep = a.thin - a.tr0 % a.thin;  // (tr0 > 0: resumed in mid-call)
  int64_t keep_off = (int64_t)(a.tr0 / a.thin) * dim;
  float eps = a.eps;

  // Energy and clamped force of the state the chain holds are CARRIED from transition to transition (as in
  // hmc_kernel.h): an accepted proposal brings its own E1 and end-of-trajectory force -- what the reference recomputes at
  // the top of the next transition on the same x, bit for bit -- a rejected one keeps the saved pair.  L evaluations per
  // transition instead of L + 1 (and, for the mixture, one exact energy instead of two).  The force is parked in LDS,
  // one slot per lane and register: [16 NT][kBlock].
  float* fpark_base = dsw_base + (kBlock / 64) * DIM;
  float* fpark = fpark_base + threadIdx.x;
  int keep = a.tr0 / a.thin;
  Tile<NT> f;
  float e_cur = 0.0f;
  if constexpr (CARRY) {
    e_cur = en.eval(a, elds, x, f, m, h);
    if constexpr (!E::kEvalGivesEnergy) e_cur = E::energy(a, elds, x, m, h);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) fpark[(16 * t + r) * kBlock] = clamp_nanprop(-f.t[t][r], -1e6f, 1e6f);
  }

  for (int tr = a.tr0; tr < a.n_mh; ++tr) {
    if (a.eps_table) eps = a.eps_table[tr];
    const float half_eps = 0.5f * eps;
    const float drift_scale = a.has_mass ? eps / a.mass_safe : eps;  // x += eps * p / max(m, 1e-10), one FMA per step

    // ---- momentum draw p ~ N(0, M)
    Tile<NT> p;
    if (a.p_noise) {
      load_rows(a.p_noise, ((int64_t)tr * a.n_chains) * dim + row, p, !SH);
    } else {
      // (the per-quad Philox counters are formed here at every transition: hoisted out of the transition loop they
      //  are 2 registers per quad held across the whole trajectory)
      uint64_t e_row = (uint64_t)chain * (uint64_t)dim - (uint64_t)lo;
      asm volatile("" : "+v"(e_row));
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k0 = 32 * t + 8 * q + 4 * h;
          const F4 n = normal4_at(a.key, (e_row + (uint64_t)k0) >> 2, a.step0 + 2ull * (uint64_t)tr);
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            if constexpr (SH) p.t[t][4 * q + i] = (k0 + i >= lo && k0 + i < hi) ? n.v[i] : 0.0f;
            else p.t[t][4 * q + i] = k0 < dim ? n.v[i] : 0.0f;  // (straight-line: see gauss_mfma.hip)
          }
        }
    }
    if (a.has_mass) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) p.t[t][r] *= a.mass_sqrt;
    }
    if constexpr (diag_mass) {
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
          const float4 mq = quad_of(mraw, t, qd);
          p.t[t][4 * qd] *= sqrtf(mq.x); p.t[t][4 * qd + 1] *= sqrtf(mq.y);
          p.t[t][4 * qd + 2] *= sqrtf(mq.z); p.t[t][4 * qd + 3] *= sqrtf(mq.w);
          if (m == 0) {  // this transition's drift factors, once per K-half
            const float4 ds = make_float4(eps / (mq.x < 1e-10f ? 1e-10f : mq.x), eps / (mq.y < 1e-10f ? 1e-10f : mq.y),
                                          eps / (mq.z < 1e-10f ? 1e-10f : mq.z), eps / (mq.w < 1e-10f ? 1e-10f : mq.w));
            *reinterpret_cast<float4*>(dsw + 32 * t + 8 * qd + 4 * h) = ds;
          }
        }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }

    float e1 = 0.0f, h0 = 0.0f, k_end = 0.0f;
    if constexpr (PW) {
      static_assert(!CARRY, "PW: nothing is carried");  // (the energy: with the force, or -- mixtures -- from E's own exact form, at the ends)
      const float k_start = kinetic(p);  // K(p0): before the first half kick
      bool bad = false;
      // E(x), and p += kick * clamp(-dE/dx) piece by piece
      // (want_e: the energy is wanted at the trajectory's two ends only -- H0, H1.  Nothing is lost for the safe mode: a position
      //  that leaves the finite range inside the trajectory never comes back -- the kicks are clamped -- and shows in the last energy,
      //  a non-finite force in the momentum and in K(p) at the end)
      auto eval_and_kick = [&](float kick, bool want_e) __attribute__((always_inline)) -> float {  // (beyond the inliner's budget at eight tiles)
        float e = 0.0f;
        gauss3::static_for<E::kPieces>([&](auto pc) {
          constexpr int T0 = E::piece_t0(decltype(pc)::value), TN = E::piece_tn(decltype(pc)::value);
          if constexpr (TN > 0) {
            f32x16 gp[TN];
            e += en.template eval_tiles<T0, TN>(a, elds, x, gp, m, h, want_e);
#pragma unroll
            for (int t = 0; t < TN; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) {
                const float fn = __builtin_amdgcn_fmed3f(-gp[t][r], -1e6f, 1e6f);
                p.t[T0 + t][r] = __builtin_fmaf(kick, fn, p.t[T0 + t][r]);
              }
          }
        });
        if (!(__builtin_fabsf(e) < __builtin_inff())) bad = true;
        return e;
      };
      e_cur = eval_and_kick(half_eps, true);
      h0 = clamp_nanprop(e_cur, -1e10f, 1e10f) + k_start;
      e1 = e_cur;
      for (int l = 0; l < a.n_leapfrog; ++l) {
        if constexpr (diag_mass) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
              const float4 ds4 = quad_of(dsw, t, qd);
              const float ds[4] = {ds4.x, ds4.y, ds4.z, ds4.w};
#pragma unroll
              for (int i = 0; i < 4; ++i) x.t[t][4 * qd + i] = __builtin_fmaf(ds[i], p.t[t][4 * qd + i], x.t[t][4 * qd + i]);
            }
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) x.t[t][r] = __builtin_fmaf(drift_scale, p.t[t][r], x.t[t][r]);
        }
        const bool last = l + 1 == a.n_leapfrog;
        e1 = eval_and_kick(last ? half_eps : eps, last);
      }
      // (a momentum that left the finite range stays outside it -- every kick is finite -- and shows in K(p) at the end; a non-finite
      //  force shows in the energy that came with it: E = (x - mu) . g / 2)
      k_end = kinetic(p);
      if (!(k_end < __builtin_inff())) bad = true;
#ifdef EBM_ABL_NOBAD
      bad = false;
#endif
      if (!vote_all<E>(!bad)) {
        // something left the fast path's domain: this workgroup (the evaluation has barriers inside: every wave of it) or this wave
        // finishes its call -- from this transition on, whose accepted state is in a.x -- in the literal body
        if constexpr (BlockVote<E>::value) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the slabs requested ahead have landed)
          __syncthreads();
        }
        GaussHmcArgs rest = a_in;
        rest.tr0 = tr;
        rest.resumed = 1;
        gauss_hmc_fallback<NT, DIAGM, E, DIAG, SH>(rest);
        return;
      }
    } else {
    // ---- H0 and the first (clamped) force: the carried pair
    if constexpr (!CARRY) {
      e_cur = en.eval(a, elds, x, f, m, h);
      if constexpr (!E::kEvalGivesEnergy) e_cur = E::energy(a, elds, x, m, h);
    }
    const float e0 = e_cur;
    h0 = clamp_nanprop(e0, -1e10f, 1e10f) + kinetic(p);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if constexpr (CARRY) f.t[t][r] = fpark[(16 * t + r) * kBlock];
        else f.t[t][r] = clamp_nanprop(-f.t[t][r], -1e6f, 1e6f);
      }

    // ---- L leapfrog steps in safe mode (see hmc_kernel.h: leapfrog_steps for the fast / literal split)
    e1 = e0;
    for (int l = 0; l < a.n_leapfrog; ++l) {
      if constexpr (diag_mass) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int qd = 0; qd < 4; ++qd) {
            const float4 ds4 = quad_of(dsw, t, qd);
            const float ds[4] = {ds4.x, ds4.y, ds4.z, ds4.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const float ph = __builtin_fmaf(half_eps, f.t[t][4 * qd + i], p.t[t][4 * qd + i]);
              p.t[t][4 * qd + i] = ph;
              x.t[t][4 * qd + i] = __builtin_fmaf(ds[i], ph, x.t[t][4 * qd + i]);
            }
          }
      } else {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float ph = __builtin_fmaf(half_eps, f.t[t][r], p.t[t][r]);
            p.t[t][r] = ph;
            x.t[t][r] = __builtin_fmaf(drift_scale, ph, x.t[t][r]);
          }
      }
      // ONE call site for the evaluation inside the step (the literal path below re-enters it with `scrubbed` set:
      // a second inlined copy of the 64 NT^2 MFMAs costs registers in the hot loop)
      bool scrubbed = false;
      for (;;) {
        e1 = en.eval(a, elds, x, f, m, h);  // f holds +g here (e1: the energy, or a finiteness witness)
        if (scrubbed) {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) f.t[t][r] = clamp_nanprop(-f.t[t][r], -1e6f, 1e6f);
          break;
        }
        // E finite => x finite, g clean.  The decision is taken per WAVE: the literal path re-runs the
        // MFMA evaluation, and an MFMA writes its result for every lane whatever EXEC says -- it must not
        // run while other chains of the wave sit in the fast path.  (For a chain that is fine the literal
        // path computes exactly what the fast path does.)
        if (vote_all<E>(__builtin_fabsf(e1) < __builtin_inff())) {
          float pz = 0.0f;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float fn = __builtin_amdgcn_fmed3f(-f.t[t][r], -1e6f, 1e6f);
              const float pn = __builtin_fmaf(half_eps, fn, p.t[t][r]);
              f.t[t][r] = fn;
              p.t[t][r] = pn;
              pz = __builtin_fmaf(pn, 0.0f, pz);
            }
          pz += __shfl_xor(pz, 32);
          if (pz != pz) {  // momentum overflow: x is finite, so f stands
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
              for (int r = 0; r < 16; ++r) p.t[t][r] = nan_to_num0(p.t[t][r]);
          }
          break;
        }
        // rare: literal semantics (NaN-propagating clamp, scrub, re-evaluate on the scrubbed x)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float fn = clamp_nanprop(-f.t[t][r], -1e6f, 1e6f);
            p.t[t][r] = nan_to_num0(__builtin_fmaf(half_eps, fn, p.t[t][r]));
            x.t[t][r] = nan_to_num0(x.t[t][r]);
          }
        scrubbed = true;
      }
    }
    }
    if constexpr (!E::kEvalGivesEnergy && !PW) e1 = E::energy(a, elds, x, m, h);
    if constexpr (!PW) k_end = kinetic(p);
    const float h1 = clamp_nanprop(e1, -1e10f, 1e10f) + k_end;

    // ---- Metropolis accept (samplers/hmc.py:277-292)
    const float dlt = clamp_nanprop(h0 - h1, -50.0f, 50.0f);
    float acc_p = expf(dlt);
    acc_p = (acc_p > 1.0f) ? 1.0f : acc_p;  // clamp_(max=1); NaN stays NaN and rejects
    float uu;
    if (a.u) uu = active ? a.u[(int64_t)tr * a.n_chains + chain] : 2.0f;
    else uu = u01_half_open(pick(philox_at(a.key, (uint64_t)chain >> 2, a.step0 + 2ull * (uint64_t)tr + 1ull), (int)(chain & 3)));
    const bool accept = active && (uu < acc_p);
    if (accept) {
      store_rows(a.x, row, x);             // the x array always holds the accepted state ...
      e_cur = e1;  // (kept for the records even when nothing is carried)
      if constexpr (CARRY) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) fpark[(16 * t + r) * kBlock] = f.t[t][r];
      }
    } else {
      load_rows(a.x, row, x);              // ... which a rejected proposal falls back to (its energy / force stay parked)
    }

    const bool leader = active && h == 0;
    if (a.accept_mask && leader) a.accept_mask[(int64_t)tr * a.n_chains + chain] = accept ? 1 : 0;
    if (a.accept_count) {
      const unsigned long long b = __ballot(accept && leader);
      if (lane == 0 && b) atomicAdd(a.accept_count + tr, (uint32_t)__popcll(b));
    }
    if ((a.traj || DIAG) && --until_keep == 0) {
      until_keep = a.thin;
      if (a.traj) store_rows(a.traj, traj_row + keep_off, x, !SH);
      keep_off += dim;
      if constexpr (DIAG) {
        // samplers/hmc.py:294-310: population mean / var, mean of the clamped energy of the state the chain holds now,
        // acceptance rate of this transition
        // (SH: the records of the K classes interleave -- record (group, class), diag.h plan_classes)
        const int64_t wave_id = SH ? ((int64_t)(blockIdx.x / (unsigned)a.sh_classes) * (kBlock / 64) + (threadIdx.x >> 6)) * a.sh_classes + sh_s
                                   : (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        diag::wave_record<NT>(a.diag.partials, a.diag.n_blocks, keep, wave_id, dim, [&](int t, int r) { return x.t[t][r]; }, active, lane, 0, lo);
        diag::wave_record_tail(a.diag.partials, a.diag.n_blocks, keep, wave_id, dim, clamp_nanprop(e_cur, -1e10f, 1e10f), active,
                               accept && leader, lane);
        ++keep;
      }
    }
  }
}

template <int NT, bool DIAGM, class E, bool DIAG, bool SH>
__device__ __noinline__ void gauss_hmc_fallback(const GaussHmcArgs& a) {
  gauss_hmc_mfma_body<NT, DIAGM, E, DIAG, SH, false>(a);
}

// dims 32 / 64 run best held to 256 VGPRs (two waves per SIMD: 0.62 vs 0.80 ms at dim 64), dims 96 / 128
// need more than that for the state alone.  (Two entry points because hipcc 7.2 silently ignores a
// template-dependent __launch_bounds__ argument.)
template <int NT, bool DIAGM, class E, bool DIAG = false, bool SH = false>
__global__ __launch_bounds__(kBlock, 2) void gauss_hmc_mfma_kernel_w2(GaussHmcArgs a) {
  gauss_hmc_mfma_body<NT, DIAGM, E, DIAG, SH, piecewise_of<E>::value>(a);
}
template <int NT, bool DIAGM, class E, bool DIAG = false, bool SH = false>
__global__ __launch_bounds__(kBlock) void gauss_hmc_mfma_kernel(GaussHmcArgs a) {
  gauss_hmc_mfma_body<NT, DIAGM, E, DIAG, SH, piecewise_of<E>::value>(a);
}
template <int NT, bool DIAGM, class E, bool DIAG = false, bool SH = false>
__global__ __launch_bounds__(kBlock, 3) void gauss_hmc_mfma_kernel_w3(GaussHmcArgs a) {
  gauss_hmc_mfma_body<NT, DIAGM, E, DIAG, SH, piecewise_of<E>::value>(a);
}

// WAVES: hold the kernel to 2 or 3 waves per SIMD (256 / 168 VGPRs); 0: unconstrained
template <int NT, bool DIAGM, class E, int WAVES, bool DIAG = false, bool SH = false>
int launch_policy(const GaussHmcArgs& a, hipStream_t st) {
  // the energy's area, raw masses, one row of drift factors per wave, the parked force (one slot per lane and register)
  const size_t smem = (size_t)(E::kLdsFloats + (1 + kBlock / 64) * 32 * NT + (E::kCarry ? 16 * NT * kBlock : 0)) * sizeof(float);
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  if (attr_once.first() && smem > 64 * 1024) {  // more than 64 KiB of dynamic LDS needs the opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_hmc_mfma_kernel<NT, DIAGM, E, DIAG, SH>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  // (SH: every class gets the workgroups of the largest one, class-major inside blockIdx: b % K)
  const int64_t blocks = SH ? ceil_div64(ceil_div64(a.n_chains, a.sh_classes), 32 * (kBlock / 64)) * a.sh_classes
                            : ceil_div64(a.n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_hmc_chain_f32: too many chains for one launch");
  if constexpr (WAVES == 3)
    hipLaunchKernelGGL((gauss_hmc_mfma_kernel_w3<NT, DIAGM, E, DIAG, SH>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else if constexpr (WAVES == 2)
    hipLaunchKernelGGL((gauss_hmc_mfma_kernel_w2<NT, DIAGM, E, DIAG, SH>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else
    hipLaunchKernelGGL((gauss_hmc_mfma_kernel<NT, DIAGM, E, DIAG, SH>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_hmc_chain_f32");
}


// the common part of the argument block
inline GaussHmcArgs matrix_hmc_args(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog,
                                    float eps, const float* eps_table, int32_t mass_kind, double mass_scalar,
                                    const float* mass_diag, int32_t thin, float* traj, uint8_t* accept_mask,
                                    uint32_t* accept_count, const float* p_noise, const float* u, uint64_t seed, uint64_t offset) {
  GaussHmcArgs a{};
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.n_mh = n_mh; a.n_leapfrog = n_leapfrog;
  a.eps = eps; a.eps_table = eps_table;
  a.has_mass = mass_kind == EBM_MASS_SCALAR;
  a.mass_raw = (float)mass_scalar;
  a.mass_sqrt = (float)sqrt(mass_scalar);
  a.mass_safe = (float)(mass_scalar < 1e-10 ? 1e-10 : mass_scalar);
  a.thin = thin; a.n_kept = n_mh / thin; a.traj = traj;
  a.accept_mask = accept_mask; a.accept_count = accept_count; a.p_noise = p_noise; a.u = u;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  a.mass_diag = mass_kind == EBM_MASS_DIAG ? mass_diag : nullptr;
  const bool mixture = e.kind == EBM_ENERGY_GMM;
  a.mean = mixture ? nullptr : e.dev0; a.prec = mixture ? nullptr : e.dev1;
  a.prec_image = (!mixture && e.kind == EBM_ENERGY_GAUSSIAN) ? reinterpret_cast<const char*>(e.aux) : nullptr;
  a.gm_means = mixture ? e.dev0 : nullptr; a.gm_logw = mixture ? e.dev1 : nullptr;
  a.n_comp = mixture ? e.n_comp : 0; a.inv2s2 = mixture ? e.s[0] : 0.0f; a.invs2 = mixture ? e.s[1] : 0.0f;
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0};
  return a;
}

}  // namespace
}  // namespace ebm
