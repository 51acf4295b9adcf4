// HMC transitions for the dense Gaussian at widths that are NOT a multiple of 4 (21 .. 157) on the matrix cores: the
// SHIFTED-row instantiations of the matrix-layout transition body (mfma_hmc_body.h SH; the idea: gauss_mfma_body.h) -- one
// alignment class of chains per workgroup, the precision matrix staged shifted by the class's offset.  A lane row is still
// ONE chain, so the Metropolis decision needs nothing new (packed rows, several chains per row, have no HMC form).
// Before: the lane-group kernel (dim 33 / 99 / 126: 0.95 / 3.9 / 90 ms per 10 transitions of 2^16 chains at L = 10).
// EBM_SHIFT_DIAG: this translation unit holds the instantiations with diagnostics records (gauss_hmc_shift_diag.hip).
// Reference: torchebm/samplers/hmc.py:243-312, core/base_model.py:181-210.
#include "mfma_hmc_body.h"

namespace ebm {

#ifndef EBM_SHIFT_DIAG
bool gauss_hmc_shift_supported(int32_t dim) { return dim >= 17 && (dim % 4) != 0 && dim + ((dim & 1) ? 3 : 2) <= 160; }
#else
bool gauss_hmc_shift_supported(int32_t dim);
#endif

namespace {
#ifdef EBM_SHIFT_DIAG
constexpr bool kRecords = true;
#else
constexpr bool kRecords = false;
#endif

template <int NT, int KT, bool DIAGM>
int launch_shift(const GaussHmcArgs& a, hipStream_t st) {
  return launch_policy<NT, DIAGM, GaussE<NT, true, KT>, 0, kRecords, true>(a, st);
}
template <bool DIAGM>
int launch_shift_nt(const GaussHmcArgs& a, hipStream_t st) {
  const int ext = a.dim + ((a.dim & 1) ? 3 : 2), nt = (ext + 31) / 32;  // tile coordinates a row can reach
  const bool trim = 32 * nt - ext >= 16;
  switch (nt) {
    case 1: return launch_shift<1, 0, DIAGM>(a, st);
    case 2: return trim ? launch_shift<2, 1, DIAGM>(a, st) : launch_shift<2, 0, DIAGM>(a, st);
    case 3: return trim ? launch_shift<3, 1, DIAGM>(a, st) : launch_shift<3, 0, DIAGM>(a, st);
    case 4: return trim ? launch_shift<4, 1, DIAGM>(a, st) : launch_shift<4, 0, DIAGM>(a, st);
    default: return trim ? launch_shift<5, 1, DIAGM>(a, st) : launch_shift<5, 0, DIAGM>(a, st);
  }
}
}  // namespace

#ifdef EBM_SHIFT_DIAG
int launch_hmc_chain_gauss_shift_diag(
#else
int launch_hmc_chain_gauss_shift(
#endif
    const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog, float eps,
    const float* eps_table, int32_t mass_kind, double mass_scalar, const float* mass_diag, int32_t thin, float* traj,
    uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise, const float* u, uint64_t seed, uint64_t offset,
    float* diag_partials, hipStream_t st) {
  if (!gauss_hmc_shift_supported(dim) || (diag_partials != nullptr) != kRecords)
    return fail(EBM_EDIM, "ebm_hmc_chain_f32: no shifted-row form for a Gaussian of dim %d", dim);
  GaussHmcArgs a = matrix_hmc_args(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag, thin,
                                   traj, accept_mask, accept_count, p_noise, u, seed, offset);
  a.sh_classes = (dim & 1) ? 4 : 2;
  if (diag_partials) {
    diag::plan_classes(n_chains, dim, a.diag);
    a.diag.partials = diag_partials;
  }
  return a.mass_diag ? launch_shift_nt<true>(a, st) : launch_shift_nt<false>(a, st);
}

}  // namespace ebm
