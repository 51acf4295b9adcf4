// HMC transitions for Gaussian mixtures at widths that are NOT a multiple of 4 (21 .. 125) on the matrix-layout transition
// body: the SHIFTED-row instantiations (mfma_hmc_body.h SH with GmmE; gmm_shift.hip for the layout).
// EBM_SHIFT_DIAG: this translation unit holds the instantiations with diagnostics records (gmm_hmc_shift_diag.hip).
// Reference: torchebm/samplers/hmc.py:243-312 over the mixture energy.
#include "mfma_hmc_body.h"

namespace ebm {

namespace {
inline int32_t shift_extent(int32_t dim) { return dim + ((dim & 1) ? 3 : 2); }
#ifdef EBM_SHIFT_DIAG
constexpr bool kRecords = true;
#else
constexpr bool kRecords = false;
#endif
}  // namespace

#ifndef EBM_SHIFT_DIAG
// plain calls: up to four tiles, three under a diagonal mass (gmm_hmc_mfma.hip); with records: three (the layout query is
// not told the mass form); up to eight components from 17 dims, more from 9 (gmm_shift.hip)
bool gmm_hmc_shift_supported(int32_t dim, int32_t n_comp, int32_t mass_kind, bool records) {
  const int max_ext = (records || mass_kind == EBM_MASS_DIAG) ? 96 : 128;
  return dim >= (n_comp > 8 ? 9 : 17) && (dim % 4) != 0 && shift_extent(dim) <= max_ext && n_comp >= 1 && n_comp <= 32;
}
#else
bool gmm_hmc_shift_supported(int32_t dim, int32_t n_comp, int32_t mass_kind, bool records);
#endif

namespace {
template <int NT, bool DIAGM>
int launch_nt(const GaussHmcArgs& a, hipStream_t st) {
  if (a.n_comp <= 8) return launch_policy<NT, DIAGM, GmmE<NT, 4>, 0, kRecords, true>(a, st);
  if (a.n_comp <= 16) return launch_policy<NT, DIAGM, GmmE<NT, 8>, 0, kRecords, true>(a, st);
  return launch_policy<NT, DIAGM, GmmE<NT, 16>, 0, kRecords, true>(a, st);
}
template <bool DIAGM>
int launch_dim(const GaussHmcArgs& a, hipStream_t st) {
  switch ((shift_extent(a.dim) + 31) / 32) {
    case 1: return launch_nt<1, DIAGM>(a, st);
    case 2: return launch_nt<2, DIAGM>(a, st);
    case 3: return launch_nt<3, DIAGM>(a, st);
    default:
      if constexpr (DIAGM || kRecords) return fail(EBM_EDIM, "ebm_hmc_chain_f32: mixture matrix kernel, diagonal mass / records: three tiles");
      else return launch_nt<4, false>(a, st);
  }
}
}  // namespace

#ifdef EBM_SHIFT_DIAG
int launch_hmc_chain_gmm_shift_diag(
#else
int launch_hmc_chain_gmm_shift(
#endif
    const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog, float eps,
    const float* eps_table, int32_t mass_kind, double mass_scalar, const float* mass_diag, int32_t thin, float* traj,
    uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise, const float* u, uint64_t seed, uint64_t offset,
    float* diag_partials, hipStream_t st) {
  if (!gmm_hmc_shift_supported(dim, e.n_comp, mass_kind, kRecords) || (diag_partials != nullptr) != kRecords)
    return fail(EBM_EDIM, "ebm_hmc_chain_f32: no shifted-row form for a mixture of dim %d", dim);
  GaussHmcArgs a = matrix_hmc_args(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag, thin,
                                   traj, accept_mask, accept_count, p_noise, u, seed, offset);
  a.sh_classes = (dim & 1) ? 4 : 2;
  if (diag_partials) {
    diag::plan_classes(n_chains, dim, a.diag);
    a.diag.partials = diag_partials;
  }
  return a.mass_diag ? launch_dim<true>(a, st) : launch_dim<false>(a, st);
}

}  // namespace ebm
