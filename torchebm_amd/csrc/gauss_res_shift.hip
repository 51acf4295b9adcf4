// Dense Gaussian Langevin chains at widths off multiples of 4 whose shifted rows reach 161 .. 256 tile coordinates (dim 158 /
// 159 .. 253 / 254): the SHIFTED-row instantiations of the register-resident streamed kernel (gauss_big_body.h,
// gauss_res_langevin_kernel SH) -- every alignment class streams its own pre-split image of the shifted precision matrix
// (ebm_gauss_prec_image_f32 writes one per class at these widths).  Before: the lane-group kernel through the C ABI, the
// GEMM step route through the sampler (dim 161 / 254, 2^16 chains x 20 steps: 2.1 / 3.5 ms where dims 160 / 256 take 0.54 / 1.27).
// Reference: torchebm/core/base_model.py:181-210 (energy), samplers/langevin_dynamics.py:154-185.
#include "gauss_big_body.h"

namespace ebm {

bool gauss_stream_shift_dim(int32_t dim);             // gauss_big_img.hip
size_t gauss_prec_image_class_bytes(int32_t dim);

bool gauss_res_shift_supported(const ebm_energy_t& e, int32_t dim) {
  return e.kind == EBM_ENERGY_GAUSSIAN && gauss_stream_shift_dim(dim) && e.aux != nullptr && (reinterpret_cast<uintptr_t>(e.aux) & 15) == 0;
}

int launch_langevin_chain_gauss_res_shift(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                                          float eta, float sqrt_eta, float noise_coef, const float* coef_table,
                                          int clamp_on, float cmin, float cmax, int32_t thin, float* traj,
                                          const float* noise, uint64_t seed, uint64_t offset, float* diag_partials, hipStream_t st) {
  if (!gauss_res_shift_supported(e, dim)) return fail(EBM_EDIM, "ebm_langevin_chain_f32: no streamed shifted-row form for a Gaussian of dim %d", dim);
  gbig::BigArgs a{};
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.noise = noise; a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj;
  a.mean = e.dev0; a.prec = e.dev1;
  a.prec_image = reinterpret_cast<const char*>(e.aux);
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  a.energy_out = nullptr; a.grad_out = nullptr;
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0};
  a.sh_classes = (dim & 1) ? 4 : 2;
  a.sh_image_stride = (int64_t)gauss_prec_image_class_bytes(dim);
  if (diag_partials) {  // one record per wave of 32 chains, the classes interleaved (diag.h plan_classes)
    diag::plan_classes(n_chains, dim, a.diag);
    a.diag.partials = diag_partials;
  }
  switch ((dim + ((dim & 1) ? 3 : 2) + 31) / 32) {
    case 6: return gbig::launch_res_shift<6>(a, st);
    case 7: return gbig::launch_res_shift<7>(a, st);
    default: return gbig::launch_res_shift<8>(a, st);
  }
}

}  // namespace ebm
