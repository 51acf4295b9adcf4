// Gaussian-mixture Langevin chains at 129 .. 256 dims on the matrix layout: five to eight tiles of the matrix-layout body
// (gauss_mfma_body.h, GKR > 0) -- the mixture's operands (one 32-row tile of components either way) still fit LDS at 256
// dims (130 KB at 32 components), the state and the gradient are 32 NT registers of a lane's 512.  Multiples of 4 as they
// are, the widths between them on shifted rows (EBM_WIDE_SH: gmm_wide_shift.hip, a translation unit of its own).
// Before: the lane-group kernels (K = 16, 2^16 chains x 20 steps: dim 129 .. 255 1.5 - 1.7 ms where dim 128 takes 0.29).
// Reference: the sampler loop of samplers/langevin_dynamics.py:154-185 over the mixture energy.
#include "gauss_mfma_body.h"

namespace ebm {
namespace {
#ifdef EBM_WIDE_SH
constexpr bool kSh = true;
#else
constexpr bool kSh = false;
#endif

template <int NT, int GKR>
__global__ __launch_bounds__(kBlock) void gmm_wide_langevin_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, false, kBlock, NT, GKR, false, 0, kSh>(a);
}
template <int NT, int GKR>
__global__ __launch_bounds__(kBlock) void gmm_wide_langevin_diag_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, false, kBlock, NT, GKR, true, 0, kSh>(a);
}

template <int NT, int GKR>
int launch_wide(const GaussArgs& a, hipStream_t st) {
  const size_t smem = (size_t)gmm3::Mixture<NT, GKR>::kLdsFloats * sizeof(float);
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  if (attr_once.first() && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_wide_langevin_kernel<NT, GKR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_wide_langevin_diag_kernel<NT, GKR>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int64_t blocks = kSh ? ceil_div64(ceil_div64(a.n_chains, a.sh_classes), 32 * (kBlock / 64)) * a.sh_classes
                             : ceil_div64(a.n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  if (a.diag.partials) hipLaunchKernelGGL((gmm_wide_langevin_diag_kernel<NT, GKR>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else hipLaunchKernelGGL((gmm_wide_langevin_kernel<NT, GKR>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_langevin_chain_f32");
}
template <int NT>
int launch_wide_nt(const GaussArgs& a, hipStream_t st) {
  if (a.gm.n_comp <= 8) return launch_wide<NT, 4>(a, st);
  if (a.gm.n_comp <= 16) return launch_wide<NT, 8>(a, st);
  return launch_wide<NT, 16>(a, st);
}
// tile coordinates a row can reach: the width itself, or (shifted rows) plus the largest class offset
inline int32_t extent(int32_t dim) { return kSh ? dim + ((dim & 1) ? 3 : 2) : dim; }
}  // namespace

#ifdef EBM_WIDE_SH
bool gmm_wide_shift_supported(int32_t dim, int32_t n_comp) {
  return (dim % 4) != 0 && extent(dim) > 128 && extent(dim) <= 256 && n_comp >= 1 && n_comp <= 32;
}
int launch_langevin_chain_gmm_wide_shift(
#else
bool gmm_wide_supported(int32_t dim, int32_t n_comp) { return (dim % 4) == 0 && dim > 128 && dim <= 256 && n_comp >= 1 && n_comp <= 32; }
int launch_langevin_chain_gmm_wide(
#endif
    const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
    const float* coef_table, int clamp_on, float cmin, float cmax, int32_t thin, float* traj, const float* noise, uint64_t seed,
    uint64_t offset, float* diag_partials, hipStream_t st) {
  GaussArgs a{};
  a.sub_dim = dim; a.pack = 1;
  a.sh_classes = kSh ? ((dim & 1) ? 4 : 2) : 1;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.mean = nullptr; a.prec = nullptr;
  a.gm = gmm3::Params{e.dev0, e.dev1, e.n_comp, dim, e.s[0], e.s[1]};
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0}; a.diag_offset_floats = 0;
  if (diag_partials) {  // one record per wave of 32 chains (shifted rows: the classes interleaved, diag.h plan_classes)
    if (kSh) diag::plan_classes(n_chains, dim, a.diag);
    else diag::plan(n_chains, dim, 32 * (int64_t)dim, a.diag);
    a.diag.partials = diag_partials;
  }
  switch ((extent(dim) + 31) / 32) {
    case 5: return launch_wide_nt<5>(a, st);
    case 6: return launch_wide_nt<6>(a, st);
    case 7: return launch_wide_nt<7>(a, st);
    case 8: return launch_wide_nt<8>(a, st);
    default: return fail(EBM_EDIM, "ebm_langevin_chain_f32: mixture matrix kernel: five to eight tiles, dim %d", dim);
  }
}

}  // namespace ebm
