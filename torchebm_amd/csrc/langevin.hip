// Langevin hot path for gfx950: the per-step Euler-Maruyama kernel and the k-fused
// chain kernel for element-wise energies (DoubleWell, Harmonic).
//
// Data layout: the chain matrix x[n_chains, dim] is fp32 row-major; for element-wise
// energies it is processed as a flat array of n_chains*dim elements, one float4 (= one
// Philox counter) per lane, so a wave64 load/store is one fully coalesced 1 KiB request.
// The k-fused kernel keeps its four elements in VGPRs for all k steps: HBM is touched
// once for the read and once for the write, plus the thinned trajectory rows.
#include <cstdlib>
#include "langevin_elem.h"

namespace ebm {

namespace {

// ---------------------------------------------------------------------------------
// per-step kernel (external gradient)
// ---------------------------------------------------------------------------------
struct StepArgs {
  const float* x;
  const float* grad;
  float* out;
  const float* noise;
  int64_t n_elem;
  StepCoef c;
  int clamp_on;
  float cmin, cmax;
  RngKey key;
  uint64_t step;
  const uint64_t* rng_dev;  // optional {seed, step} in device memory (HIP-graph replays): overrides key/step
};

__device__ __forceinline__ void resolve_rng(const StepArgs& a, RngKey& key, uint64_t& step) {
  key = a.key;
  step = a.step;
  if (a.rng_dev) {  // wave-uniform scalar loads
    const uint64_t seed = a.rng_dev[0];
    key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
    step = a.rng_dev[1];
  }
}

template <bool NOISE_PTR>
__global__ __launch_bounds__(kBlock) void langevin_step_kernel(StepArgs a) {
  const int64_t n_groups = ceil_div64(a.n_elem, 4);
  const bool draw = a.c.noise_coef != 0.0f;
  RngKey key;
  uint64_t step;
  resolve_rng(a, key, step);
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < n_groups;
       g += (int64_t)gridDim.x * kBlock) {
    const int64_t e0 = g * 4;
    const int64_t left = a.n_elem - e0;
    const int nv = left >= 4 ? 4 : (int)left;
    const F4 x = load4(a.x, e0, nv, true);
    F4 gr;
    if (a.grad) gr = load4(a.grad, e0, nv, true);
    else gr = F4{{0.f, 0.f, 0.f, 0.f}};
    F4 eps = F4{{0.f, 0.f, 0.f, 0.f}};
    if (draw) {
      if constexpr (NOISE_PTR) eps = load4(a.noise, e0, nv, true);
      else eps = normal4_at(key, (uint64_t)g, step);
    }
    F4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = draw ? em_update(x.v[i], gr.v[i], eps.v[i], a.c) : (x.v[i] - a.c.eta * gr.v[i]);
      if (a.clamp_on) v = clamp_nanprop(v, a.cmin, a.cmax);
      o.v[i] = v;
    }
    store4(a.out, e0, nv, true, o);
  }
}

// U independent float4 groups per lane per trip: all loads of a trip are issued before any of the
// RNG work, which keeps U x 32 B per lane in flight (the kernel is HBM-bound: 12 B per element).
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NOISE_PTR, int U>
__global__ __launch_bounds__(kBlock) void langevin_step_wide_kernel(StepArgs a) {
  const int64_t n_full = a.n_elem / 4;  // whole float4 groups; the launcher sends ragged tails to the plain kernel
  const bool draw = a.c.noise_coef != 0.0f;
  RngKey key;
  uint64_t step;
  resolve_rng(a, key, step);
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  for (int64_t g0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; g0 < n_full; g0 += stride * U) {
    v4f xv[U], gv[U], nv[U];
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int64_t g = g0 + j * stride;
      if (g < n_full) {
        xv[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a.x) + g);
        gv[j] = a.grad ? __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a.grad) + g) : v4f{0.f, 0.f, 0.f, 0.f};
        if constexpr (NOISE_PTR) nv[j] = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(a.noise) + g);
      }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      const int64_t g = g0 + j * stride;
      if (g >= n_full) continue;
      F4 eps = F4{{0.f, 0.f, 0.f, 0.f}};
      if (draw) {
        if constexpr (NOISE_PTR) eps = F4{{nv[j].x, nv[j].y, nv[j].z, nv[j].w}};
        else eps = normal4_at(key, (uint64_t)g, step);
      }
      const float xin[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
      const float gin[4] = {gv[j].x, gv[j].y, gv[j].z, gv[j].w};
      float o[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float v = draw ? em_update(xin[i], gin[i], eps.v[i], a.c) : (xin[i] - a.c.eta * gin[i]);
        if (a.clamp_on) v = clamp_nanprop(v, a.cmin, a.cmax);
        o[i] = v;
      }
      __builtin_nontemporal_store(v4f{o[0], o[1], o[2], o[3]}, reinterpret_cast<v4f*>(a.out) + g);
    }
  }
}

// HEUN: the reference's Heun tableau (integrators/heun.py: a = ((), (1,)), b = (1/2, 1/2)) in its
// op order (base_integrator.py:387-397): k0 = -g(x); x1 = x + h*(1*k0); k1 = -g(x1);
// x <- x + h*(0.5*k0 + 0.5*k1) + noise.  The halves are exact, the sum rounds once, so the update is
// em_update() applied to the averaged gradient.
template <int KIND, bool NOISE_PTR, bool HEUN>
__global__ __launch_bounds__(kBlock) void langevin_chain_elem_kernel(ChainArgs a) {
  const int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  const int64_t e0 = g * 4;
  if (e0 >= a.n_elem) return;
  const int64_t left = a.n_elem - e0;
  const int nv = left >= 4 ? 4 : (int)left;

  F4 x = load4(a.x, e0, nv, true);

  // trajectory addressing: traj[c, j, d] with flat e = c*dim + d
  const bool traj_vec = (a.dim & 3) == 0;
  int64_t tbase[4];
  if (a.traj) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int64_t e = e0 + i;
      const int64_t c = e / a.dim;
      const int64_t d = e - c * a.dim;
      tbase[i] = c * (int64_t)a.n_kept * a.dim + d;
    }
  }
  const bool noise_vec = (a.n_elem & 3) == 0;

  StepCoef c = a.c;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  for (int i = 0; i < a.k_steps; ++i) {
    if (a.table) {  // wave-uniform: scalar loads
      const float4 t = a.table[i];
      c.eta = t.x; c.sqrt_eta = t.y; c.noise_coef = t.z;
    }
    F4 eps;
    if constexpr (NOISE_PTR) eps = load4(a.noise + (int64_t)i * a.n_elem, e0, nv, noise_vec);
    else eps = normal4_at(a.key, (uint64_t)g, a.step0 + (uint64_t)i);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float gr = elem_grad<KIND>(x.v[j], a.s0, a.s1);
      if constexpr (HEUN) {
        const float x1 = x.v[j] - c.eta * gr;
        gr = 0.5f * gr + 0.5f * elem_grad<KIND>(x1, a.s0, a.s1);
      }
      float v = em_update(x.v[j], gr, eps.v[j], c);
      if (a.clamp_on) v = clamp_nanprop(v, a.cmin, a.cmax);
      x.v[j] = v;
    }
    if (a.traj) {
      if (--until_keep == 0) {
        until_keep = a.thin;
        if (traj_vec && nv == 4) {
          *reinterpret_cast<float4*>(a.traj + tbase[0] + keep_off) =
              make_float4(x.v[0], x.v[1], x.v[2], x.v[3]);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (j < nv) a.traj[tbase[j] + keep_off] = x.v[j];
        }
        keep_off += a.dim;
      }
    }
  }
  store4(a.x, e0, nv, true, x);
}

int grid_for(int64_t n_threads, int max_blocks) {
  int64_t b = ceil_div64(n_threads, kBlock);
  if (b < 1) b = 1;
  if (max_blocks > 0 && b > max_blocks) b = max_blocks;
  return (int)b;
}

}  // namespace

// ---------------------------------------------------------------------------------
// per-step kernel with a TENSOR diffusion coefficient (core/base_integrator.py:652-671, 724-729):
//   x' = (x - eta * grad) + sqrt(2 D_e) * (eps * sqrt_eta),   D_e = diffusion[e % period]
// (2.0 * D) ** 0.5 is torch's pow-with-0.5 = a correctly rounded fp32 square root of the rounded product.
// period = 1 (a 0-dim tensor), dim (one value per coordinate) or n_elem (a full field).
// ---------------------------------------------------------------------------------
template <bool NOISE_PTR>
__global__ __launch_bounds__(kBlock) void langevin_step_diffusion_kernel(StepArgs a, const float* __restrict__ diffusion,
                                                                         int64_t period) {
  const int64_t n_groups = ceil_div64(a.n_elem, 4);
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < n_groups; g += (int64_t)gridDim.x * kBlock) {
    const int64_t e0 = g * 4;
    const int64_t left = a.n_elem - e0;
    const int nv = left >= 4 ? 4 : (int)left;
    const F4 x = load4(a.x, e0, nv, true);
    F4 gr;
    if (a.grad) gr = load4(a.grad, e0, nv, true);
    else gr = F4{{0.f, 0.f, 0.f, 0.f}};
    F4 eps;
    if constexpr (NOISE_PTR) eps = load4(a.noise, e0, nv, true);
    else eps = normal4_at(a.key, (uint64_t)g, a.step);
    F4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float d = (i < nv) ? diffusion[period == 1 ? 0 : (e0 + i) % period] : 0.0f;
      const float coef = sqrtf(2.0f * d);
      const float x1 = x.v[i] - a.c.eta * gr.v[i];
      const float dw = eps.v[i] * a.c.sqrt_eta;
      o.v[i] = x1 + coef * dw;
    }
    store4(a.out, e0, nv, true, o);
  }
}

// ---------------------------------------------------------------------------------
// host entry points (called from api.hip)
// ---------------------------------------------------------------------------------
int launch_langevin_step_diffusion(const float* x, const float* grad, float* out, const float* noise, const float* diffusion,
                                   int64_t period, int64_t n_elem, float eta, float sqrt_eta, uint64_t seed, uint64_t offset,
                                   hipStream_t st) {
  StepArgs a{};
  a.rng_dev = nullptr;
  a.x = x; a.grad = grad; a.out = out; a.noise = noise; a.n_elem = n_elem;
  a.c = StepCoef{eta, sqrt_eta, 0.0f};
  a.clamp_on = 0; a.cmin = 0.0f; a.cmax = 0.0f;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step = offset;
  const int grid = grid_for(ceil_div64(n_elem, 4), 256 * 8);
  if (noise) hipLaunchKernelGGL(langevin_step_diffusion_kernel<true>, dim3(grid), dim3(kBlock), 0, st, a, diffusion, period);
  else hipLaunchKernelGGL(langevin_step_diffusion_kernel<false>, dim3(grid), dim3(kBlock), 0, st, a, diffusion, period);
  return check_launch("ebm_langevin_step_diffusion_f32");
}

int launch_langevin_step(const float* x, const float* grad, float* out, const float* noise,
                         int64_t n_elem, float eta, float sqrt_eta, float noise_coef, int clamp_on,
                         float cmin, float cmax, uint64_t seed, uint64_t offset, const uint64_t* rng_dev,
                         hipStream_t st) {
  StepArgs a{};
  a.rng_dev = rng_dev;
  a.x = x; a.grad = grad; a.out = out; a.noise = noise; a.n_elem = n_elem;
  a.c = StepCoef{eta, sqrt_eta, noise_coef};
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step = offset;
  // memory-bound streaming op: cap the grid at 256 CUs x 8 blocks and grid-stride the rest
  const int grid = grid_for(ceil_div64(n_elem, 4), 256 * 8);
  if ((n_elem & 3) == 0) {
    // whole float4 groups: streaming (non-temporal) loads and stores -- x, grad and out are each touched
    // exactly once, and bypassing the cache allocation is worth 20 % here (5.1 -> 6.3 TB/s at 2^26
    // elements, i.e. the chip's measured copy ceiling)
    if (noise) hipLaunchKernelGGL((langevin_step_wide_kernel<true, 1>), dim3(grid), dim3(kBlock), 0, st, a);
    else hipLaunchKernelGGL((langevin_step_wide_kernel<false, 1>), dim3(grid), dim3(kBlock), 0, st, a);
    return check_launch("ebm_langevin_step_f32");
  }
  if (noise) hipLaunchKernelGGL(langevin_step_kernel<true>, dim3(grid), dim3(kBlock), 0, st, a);
  else hipLaunchKernelGGL(langevin_step_kernel<false>, dim3(grid), dim3(kBlock), 0, st, a);
  return check_launch("ebm_langevin_step_f32");
}

int launch_langevin_chain_elem(int kind, float s0, float s1, float* x, int64_t n_chains, int32_t dim,
                               int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                               const float* coef_table, int clamp_on, float cmin, float cmax,
                               int32_t thin, float* traj, const float* noise, uint64_t seed,
                               uint64_t offset, int heun, int contracted, hipStream_t st) {
  ChainArgs a{};
  a.x = x; a.n_elem = n_chains * (int64_t)dim; a.dim = dim; a.k_steps = k_steps;
  a.c = StepCoef{eta, sqrt_eta, noise_coef};
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.s0 = s0; a.s1 = s1;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0};
  const int64_t n_groups = ceil_div64(a.n_elem, 4);
  const int64_t blocks = ceil_div64(n_groups, kBlock);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "state too large for one launch (%lld blocks)", (long long)blocks);
  const dim3 grid((unsigned)blocks), block(kBlock);
  if (!noise && heun && !traj && !coef_table && !clamp_on) {  // plain Heun chain: the lean loop with a second gradient
    if (kind == EBM_ENERGY_DOUBLE_WELL)
      hipLaunchKernelGGL((langevin_chain_lean_kernel<EBM_ENERGY_DOUBLE_WELL, false, false, false, true>), grid, block, 0, st, a);
    else
      hipLaunchKernelGGL((langevin_chain_lean_kernel<EBM_ENERGY_HARMONIC, false, false, false, true>), grid, block, 0, st, a);
    return check_launch("ebm_langevin_heun_chain_f32");
  }
  // EBM_CHAIN_CONTRACTED: a permission, used where the contracted kernel exists -- the plain call (constant coefficients, no clamp,
  // no trajectory, the kernels' own draws); every other call computes the reference's arithmetic as before
  if (contracted && !noise && !heun && !traj && !coef_table && !clamp_on) {
    if (kind == EBM_ENERGY_DOUBLE_WELL) hipLaunchKernelGGL((langevin_chain_lean_contracted_kernel<EBM_ENERGY_DOUBLE_WELL>), grid, block, 0, st, a);
    else hipLaunchKernelGGL((langevin_chain_lean_contracted_kernel<EBM_ENERGY_HARMONIC>), grid, block, 0, st, a);
    return check_launch("ebm_langevin_chain_f32");
  }
  if (!noise && !heun && (!traj || (dim & 3) == 0)) {
#define EBM_LEAN_T(KIND, TB, CL)                                                                          \
  do {                                                                                                   \
    if (traj) hipLaunchKernelGGL((langevin_chain_lean_kernel<KIND, TB, CL, true>), grid, block, 0, st, a);   \
    else hipLaunchKernelGGL((langevin_chain_lean_kernel<KIND, TB, CL, false>), grid, block, 0, st, a);       \
  } while (0)
#define EBM_LEAN(KIND)                                                 \
  do {                                                                 \
    if (coef_table && clamp_on) EBM_LEAN_T(KIND, true, true);          \
    else if (coef_table) EBM_LEAN_T(KIND, true, false);                \
    else if (clamp_on) EBM_LEAN_T(KIND, false, true);                  \
    else EBM_LEAN_T(KIND, false, false);                               \
  } while (0)
    if (kind == EBM_ENERGY_DOUBLE_WELL) EBM_LEAN(EBM_ENERGY_DOUBLE_WELL);
    else EBM_LEAN(EBM_ENERGY_HARMONIC);
#undef EBM_LEAN
#undef EBM_LEAN_T
    return check_launch("ebm_langevin_chain_f32");
  }
#define EBM_LAUNCH(KIND)                                                                         \
  do {                                                                                           \
    if (heun && noise) hipLaunchKernelGGL((langevin_chain_elem_kernel<KIND, true, true>), grid, block, 0, st, a);    \
    else if (heun) hipLaunchKernelGGL((langevin_chain_elem_kernel<KIND, false, true>), grid, block, 0, st, a);       \
    else if (noise) hipLaunchKernelGGL((langevin_chain_elem_kernel<KIND, true, false>), grid, block, 0, st, a);      \
    else hipLaunchKernelGGL((langevin_chain_elem_kernel<KIND, false, false>), grid, block, 0, st, a);                \
  } while (0)
  if (kind == EBM_ENERGY_DOUBLE_WELL) EBM_LAUNCH(EBM_ENERGY_DOUBLE_WELL);
  else EBM_LAUNCH(EBM_ENERGY_HARMONIC);
#undef EBM_LAUNCH
  return check_launch(heun ? "ebm_langevin_heun_chain_f32" : "ebm_langevin_chain_f32");
}

}  // namespace ebm
