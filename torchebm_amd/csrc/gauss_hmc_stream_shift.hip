// HMC transitions for dense Gaussians at widths off multiples of 4 whose shifted rows reach 161 .. 256 tile coordinates (dim
// 158 / 159 .. 253 / 254): the SHIFTED-row transition body (mfma_hmc_body.h SH) with the evaluation that streams the pre-split
// precision image (gauss_stream_e.h) -- every alignment class reads its own image of the shifted matrix
// (ebm_gauss_prec_image_f32 writes one per class at these widths).  Before: the per-transition GEMM route (dim 161 / 255,
// 2^16 chains, 4 transitions of 10 leapfrog steps: 11.2 / 18.5 ms where dims 160 / 256 take 1.5 / 4.0).
// EBM_SHIFT_DIAG: the instantiations with diagnostics records (gauss_hmc_stream_shift_diag.hip).
// Reference: torchebm/samplers/hmc.py:243-312, core/base_model.py:181-210.
#include "gauss_stream_e.h"

namespace ebm {

bool gauss_stream_shift_dim(int32_t dim);             // gauss_big_img.hip
size_t gauss_prec_image_class_bytes(int32_t dim);

#ifndef EBM_SHIFT_DIAG
bool gauss_hmc_stream_shift_supported(const ebm_energy_t& e, int32_t dim) {
  return e.kind == EBM_ENERGY_GAUSSIAN && gauss_stream_shift_dim(dim) && e.aux != nullptr && (reinterpret_cast<uintptr_t>(e.aux) & 15) == 0;
}
#else
bool gauss_hmc_stream_shift_supported(const ebm_energy_t& e, int32_t dim);
#endif

namespace {
#ifdef EBM_SHIFT_DIAG
constexpr bool kRecords = true;
#else
constexpr bool kRecords = false;
#endif
template <int NT, bool DIAGM>
int launch_stream_shift(const GaussHmcArgs& a, hipStream_t st) {
  return launch_policy<NT, DIAGM, GaussStreamE<NT>, 0, kRecords, true>(a, st);
}
}  // namespace

#ifdef EBM_SHIFT_DIAG
int launch_hmc_chain_gauss_stream_shift_diag(
#else
int launch_hmc_chain_gauss_stream_shift(
#endif
    const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog, float eps,
    const float* eps_table, int32_t mass_kind, double mass_scalar, const float* mass_diag, int32_t thin, float* traj,
    uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise, const float* u, uint64_t seed, uint64_t offset,
    float* diag_partials, hipStream_t st) {
  if (!gauss_hmc_stream_shift_supported(e, dim) || (diag_partials != nullptr) != kRecords)
    return fail(EBM_EDIM, "ebm_hmc_chain_f32: no streamed shifted-row form for a Gaussian of dim %d", dim);
  GaussHmcArgs a = matrix_hmc_args(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag, thin,
                                   traj, accept_mask, accept_count, p_noise, u, seed, offset);
  a.sh_classes = (dim & 1) ? 4 : 2;
  a.sh_image_stride = (int64_t)gauss_prec_image_class_bytes(dim);
  if (diag_partials) {
    diag::plan_classes(n_chains, dim, a.diag);
    a.diag.partials = diag_partials;
  }
  const int nt = (dim + ((dim & 1) ? 3 : 2) + 31) / 32;
  if (a.mass_diag) return nt == 6 ? launch_stream_shift<6, true>(a, st) : (nt == 7 ? launch_stream_shift<7, true>(a, st) : launch_stream_shift<8, true>(a, st));
  return nt == 6 ? launch_stream_shift<6, false>(a, st) : (nt == 7 ? launch_stream_shift<7, false>(a, st) : launch_stream_shift<8, false>(a, st));
}

}  // namespace ebm
