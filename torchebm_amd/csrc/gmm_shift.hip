// Gaussian-mixture Langevin chains at widths that are NOT a multiple of 4 (21 .. 125) on the matrix layout: the SHIFTED-row
// instantiations of the matrix-layout body (gauss_mfma_body.h SH, GKR > 0) -- one alignment class of chains per workgroup,
// the means staged shifted by the class's offset, the row's own padding held at 0.  Before: the lane-group kernels (2 - 4x
// the time of the neighbouring multiple of 4: dim 65 / 99 at K = 8: 0.55 / 0.57 ms per 20 steps of 2^16 chains, dim 64 / 100: 0.14 / 0.26).
// Reference: the sampler loop of samplers/langevin_dynamics.py:154-185 over the mixture energy.
#include "gauss_mfma_body.h"

namespace ebm {
namespace {

template <int NT, int GKR>
__global__ __launch_bounds__(kBlock) void gmm_shift_langevin_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, false, kBlock, NT, GKR, false, 0, true>(a);
}
template <int NT, int GKR>
__global__ __launch_bounds__(kBlock) void gmm_shift_langevin_fast_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, true, kBlock, NT, GKR, false, 0, true>(a);
}
template <int NT, int GKR>
__global__ __launch_bounds__(kBlock) void gmm_shift_langevin_diag_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, false, kBlock, NT, GKR, true, 0, true>(a);
}

template <int NT, int GKR>
int launch_gmm_shift(const GaussArgs& a, hipStream_t st) {
  const size_t smem = (size_t)gmm3::Mixture<NT, GKR>::kLdsFloats * sizeof(float);
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  if (attr_once.first() && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_shift_langevin_kernel<NT, GKR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_shift_langevin_fast_kernel<NT, GKR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_shift_langevin_diag_kernel<NT, GKR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int64_t blocks = ceil_div64(ceil_div64(a.n_chains, a.sh_classes), 32 * (kBlock / 64)) * a.sh_classes;
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  if (a.diag.partials)
    hipLaunchKernelGGL((gmm_shift_langevin_diag_kernel<NT, GKR>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else if (NT != 3 && !a.noise && !a.clamp_on)  // (three tiles: the staged form drops to one wave per SIMD and stays off -- gauss_mfma.hip)
    hipLaunchKernelGGL((gmm_shift_langevin_fast_kernel<NT, GKR>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else
    hipLaunchKernelGGL((gmm_shift_langevin_kernel<NT, GKR>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_langevin_chain_f32");
}
template <int NT>
int launch_gmm_shift_nt(const GaussArgs& a, hipStream_t st) {
  if (a.gm.n_comp <= 8) return launch_gmm_shift<NT, 4>(a, st);
  if (a.gm.n_comp <= 16) return launch_gmm_shift<NT, 8>(a, st);
  return launch_gmm_shift<NT, 16>(a, st);
}
inline int32_t shift_extent(int32_t dim) { return dim + ((dim & 1) ? 3 : 2); }  // tile coordinates a row can reach
}  // namespace

bool gmm_shift_supported(int32_t dim, int32_t n_comp) {
  // (up to eight components: from 17 dims -- below, the lane-group kernels' groups of 1 .. 4 lanes are as fast; more: from 9.
  //  K = 8, 2^16 chains x 20 steps, dims 19 / 21 / 29: 0.19 ms on the lane-group kernel, 0.11 here)
  return dim >= (n_comp > 8 ? 9 : 17) && (dim % 4) != 0 && shift_extent(dim) <= 128 && n_comp >= 1 && n_comp <= 32;
}

int launch_langevin_chain_gmm_shift(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                                    float eta, float sqrt_eta, float noise_coef, const float* coef_table,
                                    int clamp_on, float cmin, float cmax, int32_t thin, float* traj,
                                    const float* noise, uint64_t seed, uint64_t offset, float* diag_partials, hipStream_t st) {
  if (!gmm_shift_supported(dim, e.n_comp)) return fail(EBM_EDIM, "ebm_langevin_chain_f32: no shifted-row form for a mixture of dim %d", dim);
  GaussArgs a{};
  a.sub_dim = dim; a.pack = 1;
  a.sh_classes = (dim & 1) ? 4 : 2;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.mean = nullptr; a.prec = nullptr;
  a.gm = gmm3::Params{e.dev0, e.dev1, e.n_comp, dim, e.s[0], e.s[1]};
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0}; a.diag_offset_floats = 0;
  if (diag_partials) {  // one record per wave, the classes interleaved (diag.h plan_classes)
    diag::plan_classes(n_chains, dim, a.diag);
    a.diag.partials = diag_partials;
  }
  switch ((shift_extent(dim) + 31) / 32) {
    case 1: return launch_gmm_shift_nt<1>(a, st);
    case 2: return launch_gmm_shift_nt<2>(a, st);
    case 3: return launch_gmm_shift_nt<3>(a, st);
    default: return launch_gmm_shift_nt<4>(a, st);
  }
}

}  // namespace ebm
