// Isotropic Gaussian mixture in the matrix layout (shared by the Langevin and the HMC matrix bodies).
#pragma once
#include "gauss_bf16x3.h"

namespace ebm {
namespace gmm3 {

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Params {
  const float* means;  // [n_comp, dim], global
  const float* logw;   // [n_comp], global
  int32_t n_comp, dim;
  float inv2s2, invs2;  // 1 / (2 sigma^2), 1 / sigma^2
  int32_t lo = 0;       // SHIFTED rows (gauss_mfma_body.h SH): the row's column 0 sits at tile coordinate lo
};

// Isotropic Gaussian mixture, up to 32 components (core/energies.py: GaussianMixtureModel):
//   E = -logsumexp_k(log w_k - |x - mu_k|^2 / (2 sigma^2)),   dE/dx = (x - sum_k r_k mu_k) / sigma^2,  r = softmax.
// The two K x dim passes of the gradient are small GEMMs and run on the bf16 matrix pipe with split operands
// (gauss_bf16x3.h), both reading their K operand straight from registers in the C/D layout:
//   1. logits^T [comp, chain] = Mu [comp, d] . x^T [d, chain]        (one 32-row tile of components, 2 NT K-blocks)
//      l_k = c_k + (x . mu_k) / sigma^2 with c_k = log w_k - |mu_k|^2 / (2 sigma^2): softmax is shift-invariant, |x|^2 drops
//      out (the gradient-only form of rows.h); lane (n, h) holds components (r & 3) + 8 (r >> 2) + 4 h in register r, so
//      the softmax is per-lane arithmetic plus two xor-32 shuffles (max, sum);
//   2. acc^T [d, chain] = Mu^T [d, comp] . w [comp, chain]            (NT tiles, one K-block per 16 components)
//      whose result lands in the state's own layout: g = (x - acc / sum) / sigma^2 is register-to-register.
// KR = live logit registers per lane: 4 (K <= 8), 8 (K <= 16), 16 (K <= 32).  The ENERGY (needed twice per transition)
// keeps the reference's difference form, on the VALU.  LDS: A1 splits, A2 splits, c[32], log w[32], means fp32 [2 KR][DIM].
template <int NT, int KR>
struct Mixture {
  static constexpr int DIM = 32 * NT, KP = 2 * KR, KBC = KR > 8 ? 2 : 1;
  static constexpr int kA1Floats = (int)(gauss3::aop_bytes_general(1, 2 * NT) / sizeof(float));
  static constexpr int kA2Floats = (int)(gauss3::aop_bytes_general(NT, KBC) / sizeof(float));
  static constexpr int kLdsFloats = kA1Floats + kA2Floats + 64 + KP * DIM;
  __device__ static __forceinline__ void stage(const Params& a, float* lds, int n_threads) {
    const int dim = a.dim, K = a.n_comp, lo = a.lo;
    const float* mu = a.means;
    // (tile coordinate d = column d - lo; outside the row: zero, like the columns beyond dim)
    const auto mu_at = [&](int comp, int d) { return (comp < K && d >= lo && d - lo < dim) ? mu[comp * dim + (d - lo)] : 0.0f; };
    gauss3::stage_split_matrix<1, 2 * NT>([&](int comp, int d) { return mu_at(comp, d); }, reinterpret_cast<__bf16*>(lds), n_threads);
    gauss3::stage_split_matrix<NT, KBC>([&](int d, int comp) { return mu_at(comp, d); }, reinterpret_cast<__bf16*>(lds + kA1Floats), n_threads);
    float* cvec = lds + kA1Floats + kA2Floats;
    float* lw = cvec + 32;
    float* mf = lw + 32;
    for (int k = threadIdx.x; k < 32; k += n_threads) {
      float nrm = 0.0f;
      if (k < K)
        for (int d = 0; d < dim; ++d) nrm = __builtin_fmaf(mu[k * dim + d], mu[k * dim + d], nrm);
      const float w = k < K ? a.logw[k] : -__builtin_inff();
      lw[k] = w;
      cvec[k] = k < K ? __builtin_fmaf(-nrm, a.inv2s2, w) : -__builtin_inff();
    }
    for (int i = threadIdx.x; i < KP * DIM; i += n_threads) {
      const int k = i / DIM, d = i - k * DIM;
      mf[i] = mu_at(k, d);
    }
  }
  // MFMAs of one gradient: 12 NT for the logits, 6 NT KBC for the weighted mean
  static constexpr int kMfmas = 12 * NT + 6 * NT * KBC;
  // gradient into g; returns the softmax sum (in [1, K] for a finite state, NaN as soon as a coordinate is not).
  // `fill(ordinal)` is called behind every MFMA (ordinals 0 .. kMfmas - 1): the caller's independent VALU work.
  template <class Fill = gauss3::NoFill>
  __device__ static __forceinline__ float grad(const Params& a, const float* lds, const f32x16 (&x)[NT], f32x16 (&g)[NT], int lane,
                                               Fill&& fill = gauss3::NoFill{}) {
    const int h = lane >> 5;
    f32x16 dot[1];
    gauss3::contract_general<1, 2 * NT, false>(reinterpret_cast<const __bf16*>(lds), nullptr, x, dot, lane, fill);
    const float* cvec = lds + kA1Floats + kA2Floats;
    float top = -__builtin_inff();
    f32x16 w[1];
#pragma unroll
    for (int q = 0; q < KR / 4; ++q) {
      const float4 c4 = *reinterpret_cast<const float4*>(cvec + 8 * q + 4 * h);
      const float cq[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w[0][4 * q + i] = __builtin_fmaf(dot[0][4 * q + i], a.invs2, cq[i]);
        top = __builtin_fmaxf(top, w[0][4 * q + i]);  // a NaN logit resurfaces in the sum
      }
    }
    top = __builtin_fmaxf(top, __shfl_xor(top, 32));
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < KR) {
        w[0][r] = __expf(w[0][r] - top);  // 0 for the padding components (c = -inf)
        sum += w[0][r];
      } else {
        w[0][r] = 0.0f;
      }
    }
    sum += __shfl_xor(sum, 32);
    // (the weighted mean accumulates in g itself and becomes the gradient in place: no third array beside x and g)
    gauss3::contract_general<NT, KBC, false>(reinterpret_cast<const __bf16*>(lds + kA1Floats), nullptr, w, g, lane,
                                             [&](auto ord) { fill(std::integral_constant<int, 12 * NT + decltype(ord)::value>{}); });
    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) g[t][r] = a.invs2 * (x[t][r] - g[t][r] * inv);
    return sum;
  }
  // The same gradient in two calls, for bodies that take the force in PIECES of output tiles (mfma_hmc_body.h PW): weights() -- the
  // logits' contraction and the softmax, once per evaluation: w (one tile) and its sum -- and grad_tiles<T0, TN>() -- the weighted
  // mean of tiles T0 .. T0 + TN - 1 and the gradient of those tiles.  Same arithmetic as grad().
  __device__ static __forceinline__ float weights(const Params& a, const float* lds, const f32x16 (&x)[NT], f32x16 (&w)[1], int lane) {
    const int h = lane >> 5;
    f32x16 dot[1];
    gauss3::contract_general<1, 2 * NT, false>(reinterpret_cast<const __bf16*>(lds), nullptr, x, dot, lane);
    const float* cvec = lds + kA1Floats + kA2Floats;
    float top = -__builtin_inff();
#pragma unroll
    for (int q = 0; q < KR / 4; ++q) {
      const float4 c4 = *reinterpret_cast<const float4*>(cvec + 8 * q + 4 * h);
      const float cq[4] = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        w[0][4 * q + i] = __builtin_fmaf(dot[0][4 * q + i], a.invs2, cq[i]);
        top = __builtin_fmaxf(top, w[0][4 * q + i]);
      }
    }
    top = __builtin_fmaxf(top, __shfl_xor(top, 32));
    float sum = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (r < KR) {
        w[0][r] = __expf(w[0][r] - top);
        sum += w[0][r];
      } else {
        w[0][r] = 0.0f;
      }
    }
    sum += __shfl_xor(sum, 32);
    return sum;
  }
  template <int T0, int TN>
  __device__ static __forceinline__ void grad_tiles(const Params& a, const float* lds, const f32x16 (&x)[NT], const f32x16 (&w)[1], float sum,
                                                    f32x16 (&g)[TN], int lane) {
    gauss3::contract_general<TN, KBC, false, gauss3::NoFill, NT, T0>(reinterpret_cast<const __bf16*>(lds + kA1Floats), nullptr, w, g, lane);
    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) g[t][r] = a.invs2 * (x[T0 + t][r] - g[t][r] * inv);
  }
  // the exact energy, difference form, online logsumexp over the components
  __device__ static __forceinline__ float energy(const Params& a, const float* lds, const f32x16 (&x)[NT], int lane) {
    const int h = lane >> 5;
    const float* lw = lds + kA1Floats + kA2Floats + 32;
    const float* mf = lw + 32;
    float run_max = -__builtin_inff(), run_sum = 0.0f;
    for (int k = 0; k < a.n_comp; ++k) {
      float d0 = 0.0f, d1 = 0.0f;
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          // (padding quads beyond dim hold no state: the Langevin body lets them drift with the noise)
          const int k0 = 32 * t + 8 * q + 4 * h;
          const bool real = k0 + 3 >= a.lo && k0 < a.lo + a.dim;  // (a.lo > 0: the row's own padding is held at 0 by the bodies)
          const float4 mq = *reinterpret_cast<const float4*>(mf + k * DIM + 32 * t + 8 * q + 4 * h);
          const float e0 = real ? x[t][4 * q] - mq.x : 0.0f, e1 = real ? x[t][4 * q + 1] - mq.y : 0.0f;
          const float e2 = real ? x[t][4 * q + 2] - mq.z : 0.0f, e3 = real ? x[t][4 * q + 3] - mq.w : 0.0f;
          d0 = __builtin_fmaf(e0, e0, d0); d1 = __builtin_fmaf(e1, e1, d1);
          d0 = __builtin_fmaf(e2, e2, d0); d1 = __builtin_fmaf(e3, e3, d1);
        }
      float dist = d0 + d1;
      dist += __shfl_xor(dist, 32);
      const float logit = __builtin_fmaf(-dist, a.inv2s2, lw[k]);
      const float new_max = logit > run_max ? logit : run_max;
      run_sum = __builtin_fmaf(run_sum, __expf(run_max - new_max), __expf(logit - new_max));
      run_max = new_max;
    }
    return -(run_max + logf(run_sum));
  }
};

}  // namespace gmm3
}  // namespace ebm
