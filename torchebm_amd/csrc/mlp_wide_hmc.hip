// HMC transitions on the wide MLP energy (hidden width 64 / 128 / 256, dim <= 128): the evaluation of
// mlp_wide_body.h -- all four contractions on the exact-f32 matrix cores, weights in LDS (or streamed from L2 at H = 256)
// -- sits ONCE inside the transition state machine of mlp.hip's 2-D kernel: mode 0: E and force at the current state,
// mode 1: after a kick + drift, mode 2: re-evaluation on a scrubbed position (safe mode's literal path), so that every
// MFMA is reached by the whole wave whatever single chains do.  The chain state, the momentum and the force
// all live in the 32x32 C/D layout of the evaluation (lane (m, h), register r of tile td = coordinate
// 32 td + row_of(r, h) of chain m): a kick and a drift are per-register operations, the gradient arrives
// where the momentum is, and only the two scalars of a chain (energy, kinetic energy) cross the two K-halves (one
// xor-32 shuffle each).  RNG coordinates as in hmc_kernel.h: momentum at step 2t, uniforms at 2t + 1 -- the same
// (seed, step, element) field as every other route.
// Reference: torchebm/samplers/hmc.py:201-315 (transition, accept), integrators/leapfrog.py:116-187 (safe-mode leapfrog).
#include "mlp_wide_hmc_body.h"

namespace ebm {

int launch_hmc_mlp_stream(const widemlp::WideHmcArgs& a, int dt, hipStream_t st, const char* who);        // mlp_stream_hmc.hip
int launch_hmc_mlp_wide_diag(const widemlp::WideHmcArgs& a, int hidden, int dt, hipStream_t st, const char* who);  // mlp_wide_hmc_diag.hip

// the shapes the transition kernel is built for: those of the chain kernel (momentum and force are 32 DT registers on top
// of the evaluation's own; H = 256 at three or four state tiles runs with 1.4-1.8 KB of scratch per lane)
bool mlp_wide_hmc_supported(int32_t hidden, int32_t dim) {
  if (dim < 1) return false;
#ifdef EBM_MLP_H256
  return (hidden == 64 || hidden == 128 || hidden == 256) && dim <= 128;
#else
  return (hidden == 64 || hidden == 128) && dim <= 128;
#endif
}

int launch_hmc_chain_mlp_wide(int32_t hidden, const float* params, float* x, int64_t n_chains, int32_t dim, int32_t n_mh,
                              int32_t n_leapfrog, float eps, const float* eps_table, int32_t mass_kind, double mass_scalar,
                              const float* mass_diag, int32_t thin, float* traj, uint8_t* accept_mask, uint32_t* accept_count,
                              const float* p_noise, const float* u, uint64_t seed, uint64_t offset, float* diag_partials,
                              const void* w1_image, hipStream_t st, const char* who) {
  widemlp::WideHmcArgs a{};
  a.w1_image = static_cast<const char*>(w1_image);
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.n_mh = n_mh; a.n_leapfrog = n_leapfrog;
  a.eps = eps; a.eps_table = eps_table; a.mass_kind = mass_kind;
  a.mass_raw = (float)mass_scalar;
  a.mass_sqrt = (float)sqrt(mass_scalar);
  a.mass_safe = (float)(mass_scalar < 1e-10 ? 1e-10 : mass_scalar);
  a.mass_diag = mass_diag; a.thin = thin; a.n_kept = n_mh / thin; a.traj = traj;
  a.accept_mask = accept_mask; a.accept_count = accept_count; a.p_noise = p_noise; a.u = u;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.params = params;
  a.diag_partials = diag_partials; a.diag_blocks = ceil_div64(n_chains, 32);
  const int dt = (dim + 31) / 32;
#define EBM_WIDE_HMC(HTV)                                               \
  switch (dt) {                                                         \
    case 1: return widemlp::launch_hmc_one<HTV, 1, false>(a, st, who);        \
    case 2: return widemlp::launch_hmc_one<HTV, 2, false>(a, st, who);        \
    case 3: return widemlp::launch_hmc_one<HTV, 3, false>(a, st, who);        \
    default: return widemlp::launch_hmc_one<HTV, 4, false>(a, st, who);       \
  }
  if (hidden != 256 && mass_kind == EBM_MASS_DIAG) return launch_hmc_mlp_wide_diag(a, hidden, dt, st, who);
  if (hidden == 64) { EBM_WIDE_HMC(2) }
  if (hidden == 128) { EBM_WIDE_HMC(4) }
#undef EBM_WIDE_HMC
#ifdef EBM_MLP_H256
  return launch_hmc_mlp_stream(a, dt, st, who);
#else
  return fail(EBM_EDIM, "%s: hidden width %d has no transition kernel in this build", who, hidden);
#endif
}

}  // namespace ebm
