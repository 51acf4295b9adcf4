// HMC transitions for Gaussian mixtures on the matrix-layout body: launchers (mfma_hmc_body.h: GmmE, gmm_bf16x3.h).
#include "mfma_hmc_body.h"

namespace ebm {

bool gmm_hmc_mfma_supported(int32_t dim, int32_t n_comp, int32_t mass_kind) {
  // up to four tiles (the mixture's contractions are narrow -- one tile of components -- so, unlike the Gaussian's, the
  // split's transients fit beside x, p and the force); a diagonal mass: three tiles (four spill 0.6 - 0.9 KB)
  const int max_dim = mass_kind == EBM_MASS_DIAG ? 96 : 128;
  return dim >= (n_comp > 8 ? 12 : 20) && dim <= max_dim && (dim % 4) == 0 && n_comp >= 1 && n_comp <= 32;
}

namespace {
template <int NT, bool DIAGM>
int launch_gmm_nt(const GaussHmcArgs& a, hipStream_t st) {
  // one tile: three waves per SIMD (168 VGPRs, ~100 B of scratch) -- the evaluation is one dependent chain (contraction,
  // softmax, contraction), and a third wave hides more of it than the spills cost: 1.53 -> 1.42 ms at K = 9, dim 32
  constexpr int W = NT == 1 ? 3 : 0;
  if (a.n_comp <= 8) return launch_policy<NT, DIAGM, GmmE<NT, 4>, W>(a, st);
  if (a.n_comp <= 16) return launch_policy<NT, DIAGM, GmmE<NT, 8>, W>(a, st);
  return launch_policy<NT, DIAGM, GmmE<NT, 16>, W>(a, st);
}
template <bool DIAGM>
int launch_gmm_dim(const GaussHmcArgs& a, hipStream_t st) {
  switch ((a.dim + 31) / 32) {
    case 1: return launch_gmm_nt<1, DIAGM>(a, st);
    case 2: return launch_gmm_nt<2, DIAGM>(a, st);
    case 3: return launch_gmm_nt<3, DIAGM>(a, st);
    default:
      if constexpr (DIAGM) return fail(EBM_EDIM, "ebm_hmc_chain_f32: mixture matrix kernel, diagonal mass: dim <= 96");
      else return launch_gmm_nt<4, false>(a, st);
  }
}
}  // namespace

int launch_hmc_chain_gmm_mfma(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh,
                              int32_t n_leapfrog, float eps, const float* eps_table, int32_t mass_kind,
                              double mass_scalar, const float* mass_diag, int32_t thin, float* traj, uint8_t* accept_mask,
                              uint32_t* accept_count, const float* p_noise, const float* u, uint64_t seed,
                              uint64_t offset, hipStream_t st) {
  const GaussHmcArgs a = matrix_hmc_args(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag,
                                         thin, traj, accept_mask, accept_count, p_noise, u, seed, offset);
  return a.mass_diag ? launch_gmm_dim<true>(a, st) : launch_gmm_dim<false>(a, st);
}

}  // namespace ebm
