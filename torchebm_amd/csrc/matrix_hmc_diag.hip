// The matrix-layout HMC kernels with in-kernel diagnostics records (mfma_hmc_body.h: DIAG; one record per wave of 32
// chains, diag::wave_record).  Dense Gaussians at every dim the kernels take (20 .. 160; 164 .. 256 with the pre-split image) and mixtures at dims 20 .. 96 -- the
// shapes that run here under EVERY mass form, which the layout query (ebm_diag_layout) is not told; a mixture beyond 96
// takes a run with records on the lane-group kernels.
#include "mfma_hmc_body.h"
#include "gauss_stream_e.h"

namespace ebm {

bool gauss_hmc_mfma_supported(int32_t dim, int32_t mass_kind);
bool gauss_hmc_stream_supported(const ebm_energy_t& e, int32_t dim);  // gauss_hmc_stream.hip

bool matrix_hmc_diag_plan(const ebm_energy_t& e, int64_t n_chains, int32_t dim, diag::DiagArgs& d) {
  const bool gauss = e.kind == EBM_ENERGY_GAUSSIAN && (gauss_hmc_mfma_supported(dim, EBM_MASS_NONE) || gauss_hmc_stream_supported(e, dim));
  const bool mix = e.kind == EBM_ENERGY_GMM && dim >= (e.n_comp > 8 ? 12 : 20) && dim <= 96 && dim % 4 == 0 && e.n_comp >= 1 && e.n_comp <= 32 &&
                   !(dim == 32 && e.n_comp <= 8);
  if (!gauss && !mix) return false;
  return diag::plan(n_chains, dim, 32 * (int64_t)dim, d);
}

namespace {
template <int NT, bool DIAGM>
int launch_gauss_diag(const GaussHmcArgs& a, hipStream_t st) {
  if constexpr (NT >= 2)
    if (32 * NT - a.dim >= 16) return launch_policy<NT, DIAGM, GaussE<NT, true, 1>, 0, true>(a, st);
  return launch_policy<NT, DIAGM, GaussE<NT, true>, 0, true>(a, st);
}
template <int NT, bool DIAGM>
int launch_gmm_diag(const GaussHmcArgs& a, hipStream_t st) {
  if (a.n_comp <= 8) return launch_policy<NT, DIAGM, GmmE<NT, 4>, 0, true>(a, st);
  if (a.n_comp <= 16) return launch_policy<NT, DIAGM, GmmE<NT, 8>, 0, true>(a, st);
  return launch_policy<NT, DIAGM, GmmE<NT, 16>, 0, true>(a, st);
}
template <bool DIAGM>
int launch_diag(const GaussHmcArgs& a, bool mixture, hipStream_t st) {
  const int nt = (a.dim + 31) / 32;
  if (mixture) return nt == 1 ? launch_gmm_diag<1, DIAGM>(a, st) : (nt == 2 ? launch_gmm_diag<2, DIAGM>(a, st) : launch_gmm_diag<3, DIAGM>(a, st));
  if (nt == 1) return launch_gauss_diag<1, DIAGM>(a, st);
  if (nt == 2) return launch_gauss_diag<2, DIAGM>(a, st);
  if (nt == 3) return launch_gauss_diag<3, DIAGM>(a, st);
  if (nt == 4) return launch_gauss_diag<4, DIAGM>(a, st);
  if (nt == 5) return launch_gauss_diag<5, DIAGM>(a, st);
  // 164 .. 256: the streamed evaluation (the plan admitted these widths only with the image at hand)
  if (nt == 6) return launch_policy<6, DIAGM, GaussStreamE<6>, 0, true>(a, st);
  if (nt == 7) return launch_policy<7, DIAGM, GaussStreamE<7>, 0, true>(a, st);
  return launch_policy<8, DIAGM, GaussStreamE<8>, 0, true>(a, st);
}
}  // namespace

int launch_hmc_chain_matrix_diag(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh,
                                 int32_t n_leapfrog, float eps, const float* eps_table, int32_t mass_kind,
                                 double mass_scalar, const float* mass_diag, int32_t thin, float* traj, uint8_t* accept_mask,
                                 uint32_t* accept_count, const float* p_noise, const float* u, uint64_t seed,
                                 uint64_t offset, float* diag_partials, hipStream_t st) {
  GaussHmcArgs a = matrix_hmc_args(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag, thin,
                                   traj, accept_mask, accept_count, p_noise, u, seed, offset);
  if (!matrix_hmc_diag_plan(e, n_chains, dim, a.diag))
    return fail(EBM_EDIM, "ebm_hmc_chain_f32: no matrix-layout diagnostics records for this energy / dim %d", dim);
  a.diag.partials = diag_partials;
  const bool mixture = e.kind == EBM_ENERGY_GMM;
  return a.mass_diag ? launch_diag<true>(a, mixture, st) : launch_diag<false>(a, mixture, st);
}

}  // namespace ebm
