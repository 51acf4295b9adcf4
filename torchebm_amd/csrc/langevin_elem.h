// Shared pieces of the element-wise Langevin chain kernels (langevin.hip: the plain variants,
// langevin_diag.hip: the variants that also emit diagnostics records): update arithmetic, gradient
// folds, vector load / store helpers, launch arguments and the lean k-fused kernel template.
#pragma once
#include "diag.h"
#include "ebm_common.h"

namespace ebm {
namespace {

constexpr int kBlock = 256;  // 4 waves: one per SIMD of a CU

typedef float v2f __attribute__((ext_vector_type(2)));

struct StepCoef {
  float eta, sqrt_eta, noise_coef;
};

// Reference arithmetic for one element (core/base_integrator.py:397,728-729), each op
// rounded separately:  x + eta*(1.0*(-g))  ==  x - fl(eta*g)  bit for bit.
__device__ __forceinline__ float em_update(float x, float g, float eps, StepCoef c) {
  const float x1 = x - c.eta * g;
  const float dw = eps * c.sqrt_eta;
  return x1 + c.noise_coef * dw;
}

// Gradients with autograd's rounding.  Autograd evaluates (h*(2u))*(2x) and
// (0.5k)*(2x); scaling by 2 is exact and commutes with rounding, so fl(fl(h*2u)*2x) == fl(fl(4h*u)*x)
// and fl(s*2x) == fl(2s*x) bit for bit (overflow included: both sides reach inf together; the
// intermediate never lies in the denormal range).  The doubled constants are wave-uniform: two
// multiplies per element instead of four (DoubleWell), one instead of two (Harmonic).
template <int KIND>
__device__ __forceinline__ float elem_grad(float x, float s0, float s1) {
  if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL) {
    const float u = x * x - s1;                // x.pow(2) - b**2
    return ((4.0f * s0) * u) * x;              // == (h*(2u)) * (2x), pow backward twice
  } else {
    return (2.0f * s0) * x;                    // == (0.5k) * (2x)
  }
}

__device__ __forceinline__ F4 load4(const float* __restrict__ p, int64_t e0, int n_valid, bool vec) {
  F4 r;
  if (vec && n_valid == 4) {
    const float4 t = *reinterpret_cast<const float4*>(p + e0);
    r.v[0] = t.x; r.v[1] = t.y; r.v[2] = t.z; r.v[3] = t.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i) r.v[i] = (i < n_valid) ? p[e0 + i] : 0.0f;
  }
  return r;
}

__device__ __forceinline__ void store4(float* __restrict__ p, int64_t e0, int n_valid, bool vec, F4 r) {
  if (vec && n_valid == 4) {
    *reinterpret_cast<float4*>(p + e0) = make_float4(r.v[0], r.v[1], r.v[2], r.v[3]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < n_valid) p[e0 + i] = r.v[i];
  }
}

// ---------------------------------------------------------------------------------
// k-fused chain kernel, element-wise energies
// ---------------------------------------------------------------------------------
struct ChainArgs {
  float* x;
  int64_t n_elem;
  int32_t dim;
  int32_t k_steps;
  StepCoef c;
  const float4* table;  // [k] or null
  int clamp_on;
  float cmin, cmax;
  int32_t thin;
  int32_t n_kept;
  float* traj;
  const float* noise;  // [k][n_elem] or null
  float s0, s1;
  RngKey key;
  uint64_t step0;
  diag::DiagArgs diag;  // per-block diagnostics records at the kept steps (DIAG kernels)
};

extern __shared__ __attribute__((aligned(16))) float elem_smem[];

// ---------------------------------------------------------------------------------
// Lean variant (native RNG): nothing but Philox + Box-Muller + gradient + update
// inside the loop.  The per-step coefficient table (schedulers, the Energy-Matching temperature
// sweep), the clamp and the thinned trajectory store are compile-time switches, so the headline case -- constant coefficients,
// no clamp -- carries neither a branch nor a live register for them.  One float4 group per lane:
// 2 or 4 independent groups per lane were measured and change nothing (8.96 / 9.05 / 8.91 ms on
// config 2; the loop is VALU-issue bound at 8 waves/SIMD either way).
// ---------------------------------------------------------------------------------
// DIAG: at every kept step the workgroup also reduces its 1024 elements to one diagnostics record (diag.h):
// the block's column sums / M2 and its energy sum are stored -- the population statistics of
// return_diagnostics=True without leaving the k-step launch.  Lanes past the end of the state stay in the
// loop (they take part in the workgroup barriers) with nothing to load or store.  The k steps are then walked
// as n_kept runs of `thin` steps (the hot inner loop is the plain one, unrolled by two).
// CONTRACT (ABI 8, EBM_CHAIN_CONTRACTED; the opt-in of `sampler.fused_arithmetic = True` on LangevinDynamics): the same step with the arithmetic
// contracted -- the gradient's x^2 - b^2 and the drift x - eta g as fused multiply-adds, the noise coefficient folded into the
// Box-Muller radius (ebm_common.h scaled_normal4_at): 8 of the loop's 76 vector instructions per float4 group fewer.  NOT the
// reference's rounding (eager torch rounds every multiply and add); the same law, moments and Philox field.
template <int KIND, bool TABLE, bool CLAMP, bool TRAJ, bool HEUN, bool DIAG, bool CONTRACT = false>
__device__ __forceinline__ void lean_body(const ChainArgs& a) {
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t e0 = g * 4;
  if constexpr (!DIAG) {
    if (e0 >= a.n_elem) return;
  }
  const int64_t left = a.n_elem - e0;
  const int nv = left >= 4 ? 4 : (left > 0 ? (int)left : 0);
  F4 x = load4(a.x, e0 < a.n_elem ? e0 : 0, nv, true);
  StepCoef c = a.c;
  // TRAJ (dim % 4 == 0 only, so a lane's float4 never straddles two chains): traj[c, j, d..d+3]
  float* tptr = nullptr;
  if constexpr (TRAJ) {
    const int64_t chain = e0 / a.dim;
    tptr = a.traj + chain * (int64_t)a.n_kept * a.dim + (e0 - chain * a.dim);
  }
  auto one_step = [&](int i) {
    if constexpr (TABLE) {  // wave-uniform: scalar loads
      const float4 t = a.table[i];
      c.eta = t.x; c.sqrt_eta = t.y; c.noise_coef = t.z;
    }
    if constexpr (CONTRACT) {
      static_assert(!TABLE && !CLAMP && !HEUN, "the contracted form exists for the plain call only");
      const F4 e = scaled_normal4_at(a.key, (uint64_t)g, a.step0 + (uint64_t)i, c.noise_coef * c.sqrt_eta);  // (uniform product: scalar unit)
#pragma unroll
      for (int q = 0; q < 4; q += 2) {
        const v2f xv = {x.v[q], x.v[q + 1]}, ev = {e.v[q], e.v[q + 1]};
        v2f gr;
        if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL) gr = ((4.0f * a.s0) * __builtin_elementwise_fma(xv, xv, v2f{-a.s1, -a.s1})) * xv;
        else gr = (2.0f * a.s0) * xv;
        const v2f nv2 = __builtin_elementwise_fma(v2f{-c.eta, -c.eta}, gr, xv) + ev;
        x.v[q] = nv2.x;
        x.v[q + 1] = nv2.y;
      }
      return;
    }
    const F4 eps = normal4_at(a.key, (uint64_t)g, a.step0 + (uint64_t)i);
    // gradient + update on explicit 2-vectors (packed-f32 instructions); written out this way because
    // the clamp's min/max would otherwise make the compiler fall back to scalar arithmetic for all of it
#pragma unroll
    for (int q = 0; q < 4; q += 2) {
      const v2f xv = {x.v[q], x.v[q + 1]}, ev = {eps.v[q], eps.v[q + 1]};
      v2f gr;
      if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL) gr = ((4.0f * a.s0) * (xv * xv - a.s1)) * xv;  // see elem_grad
      else gr = (2.0f * a.s0) * xv;
      if constexpr (HEUN) {  // predictor x - eta*g0, corrector gradient 0.5*g0 + 0.5*g(predictor)
        const v2f xp = xv - c.eta * gr;
        v2f g1;
        if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL) g1 = ((4.0f * a.s0) * (xp * xp - a.s1)) * xp;
        else g1 = (2.0f * a.s0) * xp;
        gr = 0.5f * gr + 0.5f * g1;
      }
      const v2f x1 = xv - c.eta * gr;
      const v2f dw = ev * c.sqrt_eta;
      v2f nv2 = x1 + c.noise_coef * dw;
      if constexpr (CLAMP) nv2 = __builtin_elementwise_minimum(__builtin_elementwise_maximum(nv2, v2f{a.cmin, a.cmin}), v2f{a.cmax, a.cmax});
      x.v[q] = nv2.x;
      x.v[q + 1] = nv2.y;
    }
  };
  if constexpr (!DIAG) {
    int until_keep = a.thin;
#pragma unroll 2  // measured: 9.00 -> 8.83 ms on config 2 (4 gives no more)
    for (int i = 0; i < a.k_steps; ++i) {
      one_step(i);
      if constexpr (TRAJ) {
        if (--until_keep == 0) {  // wave-uniform
          until_keep = a.thin;
          *reinterpret_cast<float4*>(tptr) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
          tptr += a.dim;
        }
      }
    }
  } else {
    const int64_t block_e0 = (int64_t)blockIdx.x * (kBlock * 4);
    const int64_t rest = a.n_elem - block_e0;
    const int L = rest >= kBlock * 4 ? kBlock * 4 : (int)rest;  // valid elements of this workgroup
    const bool fast = diag::fast_flat_ok(a.dim);
    const int rows_b = L / a.dim;
    const float inv_rows = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(rows_b > 0 ? 1.0f / (float)rows_b : 0.0f)));
    int i = 0;
    for (int keep = 0; keep < a.n_kept; ++keep) {
      const int stop = i + a.thin;
#pragma unroll 2
      for (; i < stop; ++i) one_step(i);
      if constexpr (TRAJ) {
        if (nv == 4) *reinterpret_cast<float4*>(tptr) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
        tptr += a.dim;
      }
      float e_part = 0.0f;  // sum over the lane's elements of the per-coordinate energy term
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        float t;
        if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL) { const float u = x.v[q] * x.v[q] - a.s1; t = u * u; }
        else t = x.v[q] * x.v[q];
        e_part += (q < nv) ? t : 0.0f;
      }
      if (fast) {
        diag::emit_flat_fast(a.diag, keep, elem_smem, a.dim, make_float4(x.v[0], x.v[1], x.v[2], x.v[3]), L, inv_rows, a.s0 * e_part);
      } else {
        *reinterpret_cast<float4*>(elem_smem + 4 * threadIdx.x) = make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
        diag::emit(a.diag, keep, elem_smem, elem_smem + a.diag.E, L, a.dim, a.s0 * e_part, 0.0f);
      }
    }
#pragma unroll 2
    for (; i < a.k_steps; ++i) one_step(i);  // the trailing k % thin steps
  }
  if (nv > 0) store4(a.x, e0, nv, true, x);
}

template <int KIND, bool TABLE, bool CLAMP, bool TRAJ, bool HEUN = false>
__global__ __launch_bounds__(kBlock) void langevin_chain_lean_kernel(ChainArgs a) {
  lean_body<KIND, TABLE, CLAMP, TRAJ, HEUN, false>(a);
}
template <int KIND>
__global__ __launch_bounds__(kBlock) void langevin_chain_lean_contracted_kernel(ChainArgs a) {
  lean_body<KIND, false, false, false, false, false, true>(a);
}

// The DIAG form, held to 64 VGPRs (8 waves per SIMD like the plain kernel: the out-of-line generic emit() would
// otherwise set the kernel's register count to 80 and cost the step loop 4 %).  A template-dependent expression in
// __launch_bounds__ is ignored by hipcc 7.2, hence the second entry point.
template <int KIND, bool TABLE, bool CLAMP, bool TRAJ, bool HEUN>
__global__ __launch_bounds__(kBlock, 8) void langevin_chain_lean_diag_kernel(ChainArgs a) {
  lean_body<KIND, TABLE, CLAMP, TRAJ, HEUN, true>(a);
}

}  // namespace
}  // namespace ebm
