// HMC transitions for Gaussian mixtures at 129 .. 255 dims on the matrix-layout transition body (mfma_hmc_body.h with GmmE at
// five to eight tiles): the mixture's operands -- one 32-row tile of components either way -- still fit LDS, position,
// momentum and force are 3 x 16 NT registers of a lane's 512.  Nothing is carried between transitions (GmmE::kCarry is off
// from five tiles: no LDS for a parked force): L + 1 evaluations per transition, as the reference.  Multiples of 4 as they
// are, the widths between them on shifted rows (EBM_WIDE_SH: gmm_hmc_wide_shift.hip, a translation unit of its own).  No
// mass vector and no records here: those calls keep the lane-group kernels.
// Where it runs is what was measured (MI355X, 2^16 chains, 4 transitions x L = 10, ms; lane-group kernel -> this one;
// scripts/bench_gmm_hmc_wide.py, scripts/sweep_cliffs.py): the kernels spill (0.7 KB of scratch per lane at five tiles,
// 1.3 - 1.8 at six, 1.8 - 2.3 at seven, 2.7 - 3.3 at eight), so
//   five tiles (129 .. 160)   K = 8: 2.1 -> 0.9     K = 16: 4.4 -> 1.2     K = 32: 2.0
//   six tiles  (161 .. 192)   K = 8: 2.2 -> 1.9-2.1 K = 16: 4.4 -> 1.9-2.3 K = 32: 4.0-5.0
//   seven      (193 .. 224)   K = 8: 2.3 -> 2.7 (stays on the lane-group kernel)   K = 16: 4.4 -> 2.7-2.9   K = 32: 4.4-5.0
//   eight      (225 .. 256)   K = 8: 2.5 -> 4.7, K = 16: 4.6 -> 5.9 (dim 256 itself: 1.6 / 2.2 -> 4.5 / 6.2): not instantiated.
// Round 6: with up to 16 components the force comes in PIECES of two tiles that are kicked into the momentum at once (GmmE::eval_tiles:
// the softmax weights once per evaluation, the weighted mean piece by piece; mfma_hmc_body.h PW) -- 639 -> 291 spilled values at
// seven tiles -- and seven tiles pay for every component count:
//   K = 8:  129 .. 192: 0.70-0.93 -> 0.63-0.90   200 / 224: 2.29 / 2.34 (lane-group) -> 1.23 / 1.21
//   K = 16: 129 .. 192: 0.84-1.16 -> 0.75-1.09   200 / 224: 1.78 / 1.80 -> 1.42 / 1.45        (K = 32: one piece, as before)
// and EIGHT tiles (225 .. 255; 451 - 485 spilled values):
//   K = 8:  228 .. 254: 2.37-2.49 (lane-group) -> 1.73-1.79      K = 16: 4.48-4.55 -> 2.02-2.10
//   K = 32 (in pieces at eight tiles only): 228 .. 254: 8.49-8.57 -> 3.69-3.78
//   (dim 256 itself stays on the lane-group kernels, at their best there: K = 8 1.61 against 1.74, K = 16 2.12 against 2.04, K = 32 3.84 / 3.84)
// Reference: torchebm/samplers/hmc.py:243-312 over the mixture energy.
#include "mfma_hmc_body.h"

namespace ebm {
namespace {
#ifdef EBM_WIDE_SH
constexpr bool kSh = true;
#else
constexpr bool kSh = false;
#endif
// tile coordinates a row can reach: the width itself, or (shifted rows) plus the largest class offset
inline int32_t extent(int32_t dim) { return kSh ? dim + ((dim & 1) ? 3 : 2) : dim; }
// five to seven tiles for every component count (measured: the tables above)
// (round 5: seven tiles from nine components only, eight never.  Round 6, the force in pieces: seven for every count, eight with up to
//  16 components -- EBM_GMM_WIDE_8: 0 off, 1 all of 225 .. 256, 2 not 256 itself, where the lane-group kernels are at their best)
#ifndef EBM_GMM_WIDE_8
#define EBM_GMM_WIDE_8 2
#endif
inline bool tiles_pay(int32_t ext, int32_t n_comp, int32_t dim) {
  if (ext <= 224) return true;
#ifndef EBM_GMM_WIDE_8_K32
#define EBM_GMM_WIDE_8_K32 1
#endif
  if (EBM_GMM_WIDE_8 == 0 || ext > 256 || (n_comp > 16 && !EBM_GMM_WIDE_8_K32)) return false;
  return EBM_GMM_WIDE_8 == 1 || dim != 256;
}

template <int NT>
int launch_nt(const GaussHmcArgs& a, hipStream_t st) {
  if (a.n_comp <= 8) return launch_policy<NT, false, GmmE<NT, 4>, 0, false, kSh>(a, st);
  if (a.n_comp <= 16) return launch_policy<NT, false, GmmE<NT, 8>, 0, false, kSh>(a, st);
  return launch_policy<NT, false, GmmE<NT, 16>, 0, false, kSh>(a, st);
}
}  // namespace

#ifdef EBM_WIDE_SH
bool gmm_hmc_wide_shift_supported(int32_t dim, int32_t n_comp, int32_t mass_kind) {
  return (dim % 4) != 0 && extent(dim) > 128 && tiles_pay(extent(dim), n_comp, dim) && n_comp >= 1 && n_comp <= 32 && mass_kind != EBM_MASS_DIAG;
}
int launch_hmc_chain_gmm_wide_shift(
#else
bool gmm_hmc_wide_supported(int32_t dim, int32_t n_comp, int32_t mass_kind) {
  return (dim % 4) == 0 && dim > 128 && tiles_pay(dim, n_comp, dim) && n_comp >= 1 && n_comp <= 32 && mass_kind != EBM_MASS_DIAG;
}
int launch_hmc_chain_gmm_wide(
#endif
    const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog, float eps,
    const float* eps_table, int32_t mass_kind, double mass_scalar, const float* mass_diag, int32_t thin, float* traj,
    uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise, const float* u, uint64_t seed, uint64_t offset,
    hipStream_t st) {
  if (mass_kind == EBM_MASS_DIAG || extent(dim) <= 128 || !tiles_pay(extent(dim), e.n_comp, dim) || ((dim % 4) != 0) != kSh)
    return fail(EBM_EDIM, "ebm_hmc_chain_f32: no wide matrix-layout form for a mixture of dim %d", dim);
  GaussHmcArgs a = matrix_hmc_args(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag, thin,
                                   traj, accept_mask, accept_count, p_noise, u, seed, offset);
  a.sh_classes = kSh ? ((dim & 1) ? 4 : 2) : 1;
  switch ((extent(dim) + 31) / 32) {
    case 5: return launch_nt<5>(a, st);
    case 6: return launch_nt<6>(a, st);
    case 7: return launch_nt<7>(a, st);
    default:  // eight tiles (tiles_pay)
      if (a.n_comp <= 8) return launch_policy<8, false, GmmE<8, 4>, 0, false, kSh>(a, st);
      if (a.n_comp <= 16) return launch_policy<8, false, GmmE<8, 8>, 0, false, kSh>(a, st);
      return launch_policy<8, false, GmmE<8, 16>, 0, false, kSh>(a, st);
  }
}

}  // namespace ebm
