// Dense Gaussian Langevin chains at widths that are NOT a multiple of 4 (17 .. 158) on the matrix cores: the SHIFTED-row
// instantiations of the matrix-layout body (gauss_mfma_body.h, SH) -- one alignment class of chains per workgroup, the
// precision matrix staged shifted by that class's offset.  The flat element order, hence the Philox field and every
// element's update, is that of the flat kernels.
// Reference: torchebm/core/base_model.py:181-210 (energy), samplers/langevin_dynamics.py:154-185.
#include "gauss_mfma_body.h"

namespace ebm {
namespace {

template <int NT, int KT>
__global__ __launch_bounds__(kBlock) void gauss_shift_langevin_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, false, kBlock, NT, 0, false, KT, true>(a);
}
template <int NT, int KT>
__global__ __launch_bounds__(kBlock) void gauss_shift_langevin_fast_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, true, kBlock, (NT >= 5 ? 4 : NT), 0, false, KT, true>(a);
}
template <int NT, int KT>
__global__ __launch_bounds__(kBlock) void gauss_shift_langevin_diag_kernel(GaussArgs a) {  // with diagnostics records
  gauss_langevin_mfma_body<NT, true, false, kBlock, NT, 0, true, KT, true>(a);
}
constexpr int kWideBlock = 512;  // (three tiles: eight waves share one LDS copy of the split matrix -- gauss_mfma.hip)
template <int NT, int HIDE, int KT>
__global__ __launch_bounds__(kWideBlock) void gauss_shift_langevin_fast_wide_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, true, kWideBlock, HIDE, 0, false, KT, true>(a);
}

template <int NT, int KT>
int launch_shift(const GaussArgs& a, hipStream_t st) {
  const size_t smem = gauss3::aop_bytes(NT) + 32 * NT * sizeof(float);
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  if (attr_once.first() && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_shift_langevin_kernel<NT, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_shift_langevin_fast_kernel<NT, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_shift_langevin_diag_kernel<NT, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if constexpr (NT == 3)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_shift_langevin_fast_wide_kernel<NT, 2, KT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const bool fast = !a.noise && !a.clamp_on && !a.diag.partials;
  const int threads = (NT == 3 && fast) ? kWideBlock : kBlock;
  // class 0 is the largest: ceil(n / S) chains; every class gets that many workgroups (class-major inside blockIdx: b % S)
  const int64_t per_class = ceil_div64(a.n_chains, a.sh_classes);
  const int64_t blocks = ceil_div64(per_class, 32 * (threads / 64)) * a.sh_classes;
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  if constexpr (NT == 3) {
    if (fast) {
      hipLaunchKernelGGL((gauss_shift_langevin_fast_wide_kernel<NT, 2, KT>), dim3((unsigned)blocks), dim3(kWideBlock), smem, st, a);
      return check_launch("ebm_langevin_chain_f32");
    }
  }
  if (a.diag.partials) hipLaunchKernelGGL((gauss_shift_langevin_diag_kernel<NT, KT>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else if (fast) hipLaunchKernelGGL((gauss_shift_langevin_fast_kernel<NT, KT>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else hipLaunchKernelGGL((gauss_shift_langevin_kernel<NT, KT>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_langevin_chain_f32");
}

// tile coordinates a row can reach: dim plus the largest class offset (3 for an odd width, 2 for dim = 2 mod 4)
inline int32_t shift_extent(int32_t dim) { return dim + ((dim & 1) ? 3 : 2); }

}  // namespace

// below 17 the packed rows stay (gauss_pack_factor: the padding of a 32-wide tile outweighs the block-diagonal waste there)
bool gauss_shift_supported(int32_t dim) { return dim >= 17 && (dim % 4) != 0 && shift_extent(dim) <= 160; }

int launch_langevin_chain_gauss_shift(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                                      float eta, float sqrt_eta, float noise_coef, const float* coef_table,
                                      int clamp_on, float cmin, float cmax, int32_t thin, float* traj,
                                      const float* noise, uint64_t seed, uint64_t offset, float* diag_partials, hipStream_t st) {
  if (!gauss_shift_supported(dim)) return fail(EBM_EDIM, "ebm_langevin_chain_f32: no shifted-row form for a Gaussian of dim %d", dim);
  GaussArgs a{};
  a.sub_dim = dim; a.pack = 1;
  a.sh_classes = (dim & 1) ? 4 : 2;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.mean = e.dev0; a.prec = e.dev1;
  a.gm = gmm3::Params{nullptr, nullptr, 0, dim, 0.0f, 0.0f};
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0}; a.diag_offset_floats = 0;
  if (diag_partials) {  // one record per wave, the classes interleaved (diag.h plan_classes)
    diag::plan_classes(n_chains, dim, a.diag);
    a.diag.partials = diag_partials;
  }
  const int ext = shift_extent(dim), nt = (ext + 31) / 32;
  const bool trim = 32 * nt - ext >= 16;
  switch (nt) {
    case 1: return launch_shift<1, 0>(a, st);
    case 2: return trim ? launch_shift<2, 1>(a, st) : launch_shift<2, 0>(a, st);
    case 3: return trim ? launch_shift<3, 1>(a, st) : launch_shift<3, 0>(a, st);
    case 4: return trim ? launch_shift<4, 1>(a, st) : launch_shift<4, 0>(a, st);
    default: return trim ? launch_shift<5, 1>(a, st) : launch_shift<5, 0>(a, st);
  }
}

}  // namespace ebm
