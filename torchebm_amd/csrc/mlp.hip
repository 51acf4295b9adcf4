// Entry points of the fused two-hidden-layer SiLU MLP energy
//     E(x) = w3 . silu(W2 silu(W1 x + b1) + b2) + b3,      x in R^dim, dim <= 128, hidden width 64 / 128 / 256
// -- forward AND input-gradient inside the kernel, so the sampler no longer round-trips through autograd (~30 launches)
// every Langevin step.  Every shape runs on the wide kernels (mlp_wide.hip, mlp_wide_hmc.hip).  Rounds 1 and 2 kept a
// kernel of its own for config 5's two-moons network (dim <= 4, H = 128: layer 1 on the VALU, the two H x H contractions
// on the exact-f32 MFMA, two waves per SIMD); on the same box the split-operand bf16 form of the wide kernels takes 0.69 ms
// where it took 0.91 (65 536 chains x 20 steps) and 3.0 ms where it took 4.4 (HMC, L = 10, 10 transitions), so it is gone.
// Reference shape: examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-31 (the energy),
// torchebm/samplers/langevin_dynamics.py:154-185 (the loop), core/base_integrator.py:711-731 (the update).
#include "diag.h"
#include "ebm_common.h"

namespace ebm {

bool mlp_wide_supported(int32_t hidden, int32_t dim);
bool mlp_wide_hmc_supported(int32_t hidden, int32_t dim);  // mlp_wide_hmc.hip
int launch_hmc_chain_mlp_wide(int32_t hidden, const float* params, float* x, int64_t n_chains, int32_t dim, int32_t n_mh,
                              int32_t n_leapfrog, float eps, const float* eps_table, int32_t mass_kind, double mass_scalar,
                              const float* mass_diag, int32_t thin, float* traj, uint8_t* accept_mask, uint32_t* accept_count,
                              const float* p_noise, const float* u, uint64_t seed, uint64_t offset, float* diag_partials,
                              const void* w1_image, hipStream_t st, const char* who);
int launch_mlp_wide(int32_t hidden, const float* params, float* x, int64_t n_chains, int32_t dim, int32_t k_steps, float eta,
                    float sqrt_eta, float noise_coef, const float* coef_table, int clamp_on, float cmin, float cmax,
                    int32_t thin, float* traj, const float* noise, uint64_t seed, uint64_t offset, float* energy_out,
                    float* grad_out, float* diag_partials, const void* w1_image, hipStream_t st, const char* who,
                    const uint64_t* rng_dev = nullptr);

namespace {

int mlp_check(const ebm_energy_t& e, int32_t dim, const char* who, bool hmc) {
  if (!e.dev0) return fail(EBM_EINVAL, "%s: packed MLP parameters pointer is NULL", who);
  if (hmc ? !mlp_wide_hmc_supported(e.n_comp, dim) : !mlp_wide_supported(e.n_comp, dim))
    return fail(EBM_EDIM, "%s: the fused MLP energy supports hidden width 64 or 128 (256 in a build made with H256=1) and 1 <= dim <= 128 (got %d, %d)", who,
                e.n_comp, dim);
  return 0;
}

}  // namespace

// Records of the matrix-layout MLP kernels (mlp_wide_body.h): a wave of 32 chains is a "block" of the record geometry.
bool mlp_diag_plan(const ebm_energy_t& e, bool hmc, int64_t n_chains, int32_t dim, diag::DiagArgs& d) {
  if (!e.dev0 || (hmc ? !mlp_wide_hmc_supported(e.n_comp, dim) : !mlp_wide_supported(e.n_comp, dim))) return false;
  return diag::plan(n_chains, dim, 32 * (int64_t)dim, d);
}

int launch_langevin_chain_mlp(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                              float eta, float sqrt_eta, float noise_coef, const float* coef_table, int clamp_on,
                              float cmin, float cmax, int32_t thin, float* traj, const float* noise, uint64_t seed,
                              uint64_t offset, float* diag_partials, hipStream_t st, const uint64_t* rng_dev) {
  const char* who = rng_dev ? "ebm_langevin_chain_dev_f32" : "ebm_langevin_chain_f32";
  if (int r = mlp_check(e, dim, who, false)) return r;
  return launch_mlp_wide(e.n_comp, e.dev0, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin, cmax,
                         thin, traj, noise, seed, offset, nullptr, nullptr, diag_partials, e.aux, st, who, rng_dev);
}

int launch_hmc_chain_mlp(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog,
                         float eps, const float* eps_table, int32_t mass_kind, double mass_scalar, const float* mass_diag,
                         int32_t thin, float* traj, uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise,
                         const float* u, uint64_t seed, uint64_t offset, float* diag_partials, hipStream_t st) {
  const char* who = "ebm_hmc_chain_f32";
  if (int r = mlp_check(e, dim, who, true)) return r;
  return launch_hmc_chain_mlp_wide(e.n_comp, e.dev0, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar,
                                   mass_diag, thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, diag_partials, e.aux, st, who);
}

int launch_energy_grad_mlp(const ebm_energy_t& e, const float* x, int64_t n_chains, int32_t dim, float* e_out,
                           float* g_out, hipStream_t st) {
  const char* who = "ebm_energy_grad_f32";
  if (int r = mlp_check(e, dim, who, false)) return r;
  return launch_mlp_wide(e.n_comp, e.dev0, const_cast<float*>(x), n_chains, dim, 0, 0.0f, 0.0f, 0.0f, nullptr, 0, 0.0f, 0.0f, 1,
                         nullptr, nullptr, 0, 0, e_out, g_out, nullptr, e.aux, st, who);
}

}  // namespace ebm
