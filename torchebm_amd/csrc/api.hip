// C-ABI entry points of libebm_hip.so (declared in include/ebm_hip.h): argument
// validation, error strings, and dispatch to the kernel launchers.  Nothing here
// allocates device memory or synchronises.
#include <cstdarg>
#include <cstdio>
#include <cstdlib>

#include "diag.h"
#include "ebm_common.h"

namespace ebm {

// launchers implemented in the kernel translation units
int launch_langevin_step(const float*, const float*, float*, const float*, int64_t, float, float,
                         float, int, float, float, uint64_t, uint64_t, const uint64_t*, hipStream_t);
int launch_langevin_step_diffusion(const float*, const float*, float*, const float*, const float*, int64_t, int64_t, float, float,
                                   uint64_t, uint64_t, hipStream_t);
int launch_langevin_chain_elem(int, float, float, float*, int64_t, int32_t, int32_t, float, float,
                               float, const float*, int, float, float, int32_t, float*,
                               const float*, uint64_t, uint64_t, int heun, int contracted, hipStream_t);
int launch_langevin_chain_rows(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float,
                               float, const float*, int, float, float, int32_t, float*,
                               const float*, uint64_t, uint64_t, int heun, float* diag_partials, hipStream_t);
int launch_langevin_chain_elem_diag(int, float, float, float*, int64_t, int32_t, int32_t, float, float, float,
                                    const float*, int, float, float, int32_t, float*, uint64_t, uint64_t, int heun,
                                    float* diag_partials, hipStream_t);
bool elem_diag_supported(int32_t dim, bool has_noise, bool has_traj);
bool elem_diag_plan(int64_t n_chains, int32_t dim, diag::DiagArgs&);
bool rows_langevin_diag_plan(const ebm_energy_t&, int heun, int64_t n_chains, int32_t dim, diag::DiagArgs&);
bool hmc_diag_plan(const ebm_energy_t&, int64_t n_chains, int32_t dim, diag::DiagArgs&);
int launch_diag_finish(const float*, int32_t, int64_t, int32_t, int32_t, int64_t, int32_t, float*, float*, float*, float*,
                       double*, hipStream_t);
int launch_hmc_chain_mlp(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t,
                         double, const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*,
                         uint64_t, uint64_t, float*, hipStream_t);
bool mlp_diag_plan(const ebm_energy_t&, bool hmc, int64_t, int32_t, diag::DiagArgs&);  // mlp.hip
int launch_hmc_chain(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float,
                     const float*, int32_t, double, const float*, int32_t, float*, uint8_t*,
                     uint32_t*, const float*, const float*, uint64_t, uint64_t, float* diag_partials, hipStream_t);
int launch_leapfrog_kick_drift(const float*, const float*, const float*, float*, float*, int64_t,
                               int32_t, float, int32_t, double, const float*, int32_t, hipStream_t);
int launch_leapfrog_kick(float*, const float*, const float*, float*, int64_t, float, int32_t,
                         hipStream_t);
int launch_hmc_accept(float*, const float*, const float*, const float*, const float*, uint8_t*,
                      uint32_t*, int64_t, int32_t, uint64_t, uint64_t, const uint64_t*, hipStream_t);
int launch_energy_grad(const ebm_energy_t&, const float*, int64_t, int32_t, float*, float*,
                       hipStream_t);
int launch_chain_stats(const float*, int64_t, int32_t, float*, float*, double*, hipStream_t);
int launch_descent_chain(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, const float*, int32_t, float,
                         int32_t, float*, hipStream_t);
int launch_descent_step(const float*, const float*, float*, float*, int64_t, float, float, hipStream_t);
int launch_lookahead(const float*, const float*, float*, int64_t, float, hipStream_t);
int launch_gmm_active_columns(const float*, int32_t, int32_t, int32_t*, hipStream_t);
int launch_hmc_chain_audit(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t, double,
                           const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*, uint64_t, uint64_t, hipStream_t);
int launch_pcd_gather(const float*, int64_t, int32_t, float*, int64_t, int64_t, const int64_t*, int64_t*, uint64_t,
                      uint64_t, const uint64_t*, hipStream_t);
int launch_pcd_scatter(float*, int64_t, int32_t, const float*, int64_t, int64_t, const int64_t*, hipStream_t);
int64_t cd_loss_work_bytes();  // misc.hip
int launch_cd_loss(const float*, int64_t, float, void*, float*, float*, hipStream_t);
int launch_cd_loss_seed(const float*, int64_t, float, const float*, const float*, float*, hipStream_t);
int launch_pcd_start_points(const float*, int64_t, int32_t, float*, int64_t, int64_t, int64_t, float, uint64_t, uint64_t, const uint64_t*,
                            hipStream_t);
int launch_noise_fill(float*, int64_t, int32_t, uint64_t, uint64_t, const uint64_t*, hipStream_t);
int launch_langevin_chain_mlp(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float, const float*,
                              int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t,
                              const uint64_t* rng_dev = nullptr);
int launch_energy_grad_mlp(const ebm_energy_t&, const float*, int64_t, int32_t, float*, float*, hipStream_t);
int launch_mlp_backward_acts(int32_t, const float*, const float*, int64_t, int32_t, const float*, float*, float*, float*, hipStream_t, const char*);
int64_t mlp_param_grads_work_floats(int32_t hidden, int32_t dim, int64_t n);  // mlp_param_grads.hip
int launch_mlp_param_grads(int32_t, const float*, const float*, int64_t, int32_t, const float*, const float*, float*, float*, hipStream_t, const char*);
int launch_probe_valu(float*, int32_t, int32_t, hipStream_t);
int launch_probe_issue(float*, int32_t, int32_t, int32_t, hipStream_t);
size_t mlp_w1_image_bytes(int32_t hidden, int32_t dim);  // mlp_wide_slab.hip
int launch_mlp_w1_image(const float* params, int32_t hidden, int32_t dim, void* image, hipStream_t st, const char* who);
size_t gauss_prec_image_bytes(int32_t dim);  // gauss_big_img.hip
int launch_gauss_prec_image(const float* prec, int32_t dim, void* image, hipStream_t st, const char* who);
bool gauss_mfma_supported(int32_t dim);
bool gauss_lds5_supported(int32_t dim);   // gauss_mfma.hip: 132 .. 160, Ps resident in LDS (plain Langevin call)
int32_t gauss_pack_factor(int32_t dim, int64_t n_chains);  // gauss_mfma.hip: 1 as is, > 1 packed rows, 0 no matrix-layout form
bool gauss_big_supported(int32_t dim);                      // gauss_big.hip: dims 132 .. 512 in steps of 4, tiled per step
int launch_langevin_chain_gauss_big(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                    const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t,
                                    float*, hipStream_t);
bool gauss_big_diag_plan(int64_t, int32_t, diag::DiagArgs&);  // gauss_big.hip: one record per wave-tile of 32 chains
int launch_energy_grad_gauss_big(const ebm_energy_t&, const float*, int64_t, int32_t, float*, float*, hipStream_t);
bool gmm_mfma_supported(int32_t dim, int32_t n_comp);
bool matrix_langevin_diag_plan(const ebm_energy_t&, int64_t, int32_t, diag::DiagArgs&);
int launch_langevin_chain_matrix_diag(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                      const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t,
                                      float*, hipStream_t);
int launch_langevin_chain_gmm_mfma(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                   const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t,
                                   hipStream_t);
bool gauss_res_shift_supported(const ebm_energy_t& e, int32_t dim);  // gauss_res_shift.hip: widths off multiples of 4 up to 254, per-class images
int launch_langevin_chain_gauss_res_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                          const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);
bool gmm_wide_supported(int32_t dim, int32_t n_comp);        // gmm_wide.hip: mixtures at 132 .. 256 dims (five to eight tiles)
bool gmm_wide_shift_supported(int32_t dim, int32_t n_comp);  // gmm_wide_shift.hip: ... and the widths off multiples of 4 between 126 and 254
int launch_langevin_chain_gmm_wide(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                   const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);
int launch_langevin_chain_gmm_wide_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                         const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);
bool gmm_shift_supported(int32_t dim, int32_t n_comp);  // gmm_shift.hip: mixtures at widths off multiples of 4, 21 .. 125
int launch_langevin_chain_gmm_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                    const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);
bool gauss_shift_supported(int32_t dim);  // gauss_shift.hip: widths off multiples of 4, 21 .. 157, on shifted rows
int launch_langevin_chain_gauss_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                      const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);
int launch_langevin_chain_gauss_mfma(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                     const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t,
                                     hipStream_t);

namespace {
thread_local char g_err[512] = "";
}

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int check_launch(const char* what) {
  const hipError_t e = hipGetLastError();
  if (e == hipSuccess) return 0;
  set_error("%s: launch failed: %s", what, hipGetErrorString(e));
  return (int)e;
}

namespace {

int check_energy(const ebm_energy_t* en, int32_t dim, const char* who) {
  if (!en) return fail(EBM_EINVAL, "%s: energy descriptor is NULL", who);
  switch (en->kind) {
    case EBM_ENERGY_DOUBLE_WELL:
    case EBM_ENERGY_HARMONIC:
      return 0;
    case EBM_ENERGY_GAUSSIAN:
      if (!en->dev0 || !en->dev1) return fail(EBM_EINVAL, "%s: Gaussian energy needs mean and precision pointers", who);
      return 0;
    case EBM_ENERGY_GMM:
      if (!en->dev0 || !en->dev1 || en->n_comp < 1) return fail(EBM_EINVAL, "%s: mixture energy needs means, log-weights and n_comp >= 1", who);
      if (en->n_comp > 64) return fail(EBM_EDIM, "%s: at most 64 mixture components are supported (got %d)", who, en->n_comp);
      return 0;
    case EBM_ENERGY_MLP:
      if (!en->dev0) return fail(EBM_EINVAL, "%s: MLP energy needs the packed parameter pointer", who);
      return 0;
    default:
      return fail(EBM_EKIND, "%s: unknown energy kind %d", who, en->kind);
  }
  (void)dim;
}

int reject_mlp(const ebm_energy_t* en, const char* who) {
  if (en->kind == EBM_ENERGY_MLP)
    return fail(EBM_EKIND, "%s: the MLP energy is fused for Langevin chains and energy/gradient evaluation only", who);
  return 0;
}

// Which kernel family serves a chain call that asks for diagnostics records, and with what record geometry.
// One function for the layout query and for the dispatch, so the two cannot disagree.
enum DiagFamily { kDiagNone = 0, kDiagElemFlat, kDiagRows, kDiagHmcRows, kDiagMatrix, kDiagMlp, kDiagGaussBig };

DiagFamily plan_diag(const ebm_energy_t& e, int sampler, int64_t n_chains, int32_t dim, bool has_noise, bool has_traj,
                     diag::DiagArgs& d) {
  if (e.kind == EBM_ENERGY_MLP)  // matrix-layout kernels: one record per wave of 32 chains (mlp_wide_body.h); no Heun kernel
    return (sampler != EBM_DIAG_LANGEVIN_HEUN && mlp_diag_plan(e, sampler == EBM_DIAG_HMC, n_chains, dim, d)) ? kDiagMlp : kDiagNone;
  const bool elementwise = e.kind == EBM_ENERGY_DOUBLE_WELL || e.kind == EBM_ENERGY_HARMONIC;
  if (sampler == EBM_DIAG_HMC) return hmc_diag_plan(e, n_chains, dim, d) ? kDiagHmcRows : kDiagNone;
  const int heun = sampler == EBM_DIAG_LANGEVIN_HEUN;
  if (elementwise && elem_diag_supported(dim, has_noise, has_traj) && elem_diag_plan(n_chains, dim, d)) return kDiagElemFlat;
  // dense Gaussians and mixtures where the matrix-layout kernels run (dims up to 96): records from those kernels
  static const bool gauss_rows = ab_switch("EBM_GAUSS_ROWS");
  static const bool gmm_rows = ab_switch("EBM_GMM_ROWS");
  const bool forced_rows = (e.kind == EBM_ENERGY_GAUSSIAN && gauss_rows) || (e.kind == EBM_ENERGY_GMM && gmm_rows);
  if (!heun && !forced_rows && matrix_langevin_diag_plan(e, n_chains, dim, d)) return kDiagMatrix;
  // dense Gaussians above 128 dims: the streamed-Ps kernels (gauss_big.hip) and their records
  if (!heun && !forced_rows && e.kind == EBM_ENERGY_GAUSSIAN && gauss_big_diag_plan(n_chains, dim, d)) return kDiagGaussBig;
  return rows_langevin_diag_plan(e, heun, n_chains, dim, d) ? kDiagRows : kDiagNone;
}

int check_state(const void* x, int64_t n_chains, int32_t dim, const char* who) {
  if (!x) return fail(EBM_EINVAL, "%s: state pointer is NULL", who);
  if (n_chains < 0 || dim < 1) return fail(EBM_EINVAL, "%s: bad shape [%lld, %d]", who, (long long)n_chains, dim);
  if (!aligned16(x)) return fail(EBM_EINVAL, "%s: state pointer must be 16-byte aligned", who);
  return 0;
}

}  // namespace
}  // namespace ebm

using namespace ebm;

extern "C" {

int ebm_version(void) { return EBM_ABI_VERSION; }

const char* ebm_last_error_string(void) { return g_err; }

int ebm_langevin_step_f32(const float* x, const float* grad, float* out, const float* noise,
                          int64_t n_elem, float eta, float sqrt_eta, float noise_coef,
                          int32_t clamp_on, float cmin, float cmax, uint64_t seed, uint64_t offset,
                          void* stream) {
  const char* who = "ebm_langevin_step_f32";
  if (n_elem < 0) return fail(EBM_EINVAL, "%s: n_elem < 0", who);
  if (n_elem == 0) return 0;
  if (!x || !out) return fail(EBM_EINVAL, "%s: x/out is NULL", who);
  if (!aligned16(x) || !aligned16(out) || (grad && !aligned16(grad)) || (noise && !aligned16(noise)))
    return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  return launch_langevin_step(x, grad, out, noise, n_elem, eta, sqrt_eta, noise_coef, clamp_on,
                              cmin, cmax, seed, offset, nullptr, (hipStream_t)stream);
}

int ebm_langevin_step_diffusion_f32(const float* x, const float* grad, float* out, const float* noise, const float* diffusion,
                                    int64_t diffusion_period, int64_t n_elem, float eta, float sqrt_eta, uint64_t seed,
                                    uint64_t offset, void* stream) {
  const char* who = "ebm_langevin_step_diffusion_f32";
  if (n_elem < 0) return fail(EBM_EINVAL, "%s: n_elem < 0", who);
  if (n_elem == 0) return 0;
  if (!x || !out || !diffusion) return fail(EBM_EINVAL, "%s: x/out/diffusion is NULL", who);
  if (diffusion_period < 1 || (diffusion_period != 1 && n_elem % diffusion_period != 0))
    return fail(EBM_EINVAL, "%s: diffusion_period %lld does not tile n_elem %lld", who, (long long)diffusion_period, (long long)n_elem);
  if (!aligned16(x) || !aligned16(out) || (grad && !aligned16(grad)) || (noise && !aligned16(noise)))
    return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  return launch_langevin_step_diffusion(x, grad, out, noise, diffusion, diffusion_period, n_elem, eta, sqrt_eta, seed, offset,
                                        (hipStream_t)stream);
}

int ebm_langevin_step_dev_f32(const float* x, const float* grad, float* out, int64_t n_elem, float eta,
                              float sqrt_eta, float noise_coef, int32_t clamp_on, float cmin, float cmax,
                              const uint64_t* rng_state, void* stream) {
  const char* who = "ebm_langevin_step_dev_f32";
  if (n_elem < 0) return fail(EBM_EINVAL, "%s: n_elem < 0", who);
  if (n_elem == 0) return 0;
  if (!x || !out || !rng_state) return fail(EBM_EINVAL, "%s: x/out/rng_state is NULL", who);
  if (!aligned16(x) || !aligned16(out) || (grad && !aligned16(grad)))
    return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  return launch_langevin_step(x, grad, out, nullptr, n_elem, eta, sqrt_eta, noise_coef, clamp_on, cmin, cmax, 0, 0,
                              rng_state, (hipStream_t)stream);
}

static int langevin_chain_impl(const char* who, int heun, const ebm_energy_t* energy, float* x, int64_t n_chains,
                               int32_t dim, int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                               const float* coef_table, int32_t clamp_on, float cmin, float cmax,
                               int32_t thin, float* traj, float* diag_partials, const float* noise, uint64_t seed,
                               uint64_t offset, void* stream) {
  if (int r = check_energy(energy, dim, who)) return r;
  if (heun) {
    if (int r = reject_mlp(energy, who)) return r;
  }
  if (int r = check_state(x, n_chains, dim, who)) return r;
  // ABI 8: `clamp_on` is a flag word -- bit 0 the clamp, bit 1 EBM_CHAIN_CONTRACTED (include/ebm_hip.h)
  if (clamp_on & ~(EBM_CHAIN_CLAMP | EBM_CHAIN_CONTRACTED)) return fail(EBM_EINVAL, "%s: unknown bits in clamp_on (%d)", who, clamp_on);
  const int contracted = (clamp_on & EBM_CHAIN_CONTRACTED) != 0;
  clamp_on &= EBM_CHAIN_CLAMP;
  if (k_steps < 0 || thin < 1) return fail(EBM_EINVAL, "%s: k_steps=%d thin=%d", who, k_steps, thin);
  if (n_chains == 0 || k_steps == 0) return 0;
  if ((coef_table && !aligned16(coef_table)) || (traj && !aligned16(traj)) || (noise && !aligned16(noise)) ||
      (diag_partials && !aligned16(diag_partials)))
    return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  if (diag_partials && k_steps / thin > 0) {
    diag::DiagArgs d;
    const DiagFamily fam = plan_diag(*energy, heun ? EBM_DIAG_LANGEVIN_HEUN : EBM_DIAG_LANGEVIN, n_chains, dim, noise != nullptr,
                                     traj != nullptr, d);
    if (fam == kDiagElemFlat)
      return launch_langevin_chain_elem_diag(energy->kind, energy->s[0], energy->s[1], x, n_chains, dim, k_steps, eta, sqrt_eta,
                                             noise_coef, coef_table, clamp_on, cmin, cmax, thin, traj, seed, offset, heun,
                                             diag_partials, (hipStream_t)stream);
    if (fam == kDiagRows)
      return launch_langevin_chain_rows(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin,
                                        cmax, thin, traj, noise, seed, offset, heun, diag_partials, (hipStream_t)stream);
    if (fam == kDiagMlp)
      return launch_langevin_chain_mlp(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin, cmax,
                                       thin, traj, noise, seed, offset, diag_partials, (hipStream_t)stream);
    if (fam == kDiagGaussBig)
      return launch_langevin_chain_gauss_big(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin,
                                             cmax, thin, traj, noise, seed, offset, diag_partials, (hipStream_t)stream);
    if (fam == kDiagMatrix)
      return launch_langevin_chain_matrix_diag(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on,
                                               cmin, cmax, thin, traj, noise, seed, offset, diag_partials, (hipStream_t)stream);
    return fail(energy->kind == EBM_ENERGY_MLP ? EBM_EKIND : EBM_EDIM,
                "%s: no in-kernel diagnostics for this energy / dim %d (see ebm_diag_layout)", who, dim);
  }
  if (energy->kind == EBM_ENERGY_MLP)
    return launch_langevin_chain_mlp(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table,
                                     clamp_on, cmin, cmax, thin, traj, noise, seed, offset, nullptr, (hipStream_t)stream);
  if (energy->kind == EBM_ENERGY_DOUBLE_WELL || energy->kind == EBM_ENERGY_HARMONIC)
    return launch_langevin_chain_elem(energy->kind, energy->s[0], energy->s[1], x, n_chains, dim,
                                      k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin,
                                      cmax, thin, traj, noise, seed, offset, heun, contracted, (hipStream_t)stream);
  if (!heun && energy->kind == EBM_ENERGY_GAUSSIAN && gauss_shift_supported(dim)) {
    // A/B switch: EBM_GAUSS_NOSHIFT=1 keeps the packed rows / the lane-group kernel for widths off multiples of 4
    static const bool no_shift = ab_switch("EBM_GAUSS_NOSHIFT");
    if (!no_shift)
      return launch_langevin_chain_gauss_shift(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table,
                                               clamp_on, cmin, cmax, thin, traj, noise, seed, offset, nullptr, (hipStream_t)stream);
  }
  if (!heun && gauss_res_shift_supported(*energy, dim)) {
    static const bool no_shift = ab_switch("EBM_GAUSS_NOSHIFT");
    if (!no_shift)
      return launch_langevin_chain_gauss_res_shift(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table,
                                                   clamp_on, cmin, cmax, thin, traj, noise, seed, offset, nullptr, (hipStream_t)stream);
  }
  if (!heun && energy->kind == EBM_ENERGY_GAUSSIAN && gauss_pack_factor(dim, n_chains) >= 1) {
    // A/B switch for tests and profiling: EBM_GAUSS_ROWS=1 keeps the LDS mat-vec kernel
    static const bool force_rows = ab_switch("EBM_GAUSS_ROWS");
    if (!force_rows)
      return launch_langevin_chain_gauss_mfma(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table,
                                              clamp_on, cmin, cmax, thin, traj, noise, seed, offset, (hipStream_t)stream);
  }
  if (!heun && energy->kind == EBM_ENERGY_GAUSSIAN && gauss_lds5_supported(dim)) {
    static const bool force_big = ab_switch("EBM_GAUSS_NO_LDS5");
    if (!force_big)
      return launch_langevin_chain_gauss_mfma(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table,
                                              clamp_on, cmin, cmax, thin, traj, noise, seed, offset, (hipStream_t)stream);
  }
  if (!heun && energy->kind == EBM_ENERGY_GAUSSIAN && gauss_big_supported(dim)) {
    static const bool force_rows = ab_switch("EBM_GAUSS_ROWS");
    if (!force_rows)
      return launch_langevin_chain_gauss_big(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table,
                                             clamp_on, cmin, cmax, thin, traj, noise, seed, offset, nullptr, (hipStream_t)stream);
  }
  if (!heun && energy->kind == EBM_ENERGY_GMM && (gmm_wide_supported(dim, energy->n_comp) || gmm_wide_shift_supported(dim, energy->n_comp))) {
    static const bool no_wide = ab_switch("EBM_GMM_NOWIDE");  // A/B switch: the lane-group kernels above 128 dims
    if (!no_wide)
      return (gmm_wide_supported(dim, energy->n_comp) ? launch_langevin_chain_gmm_wide : launch_langevin_chain_gmm_wide_shift)(
          *energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin, cmax, thin, traj, noise, seed, offset,
          nullptr, (hipStream_t)stream);
  }
  if (!heun && energy->kind == EBM_ENERGY_GMM && gmm_shift_supported(dim, energy->n_comp)) {
    static const bool no_shift = ab_switch("EBM_GAUSS_NOSHIFT");
    if (!no_shift)
      return launch_langevin_chain_gmm_shift(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table,
                                             clamp_on, cmin, cmax, thin, traj, noise, seed, offset, nullptr, (hipStream_t)stream);
  }
  // mixtures of up to 32 components on the matrix layout (gauss_mfma.hip / gmm_bf16x3.h); dims 16 / 32 with K <= 8 keep
  // one lane per chain with the means as scalar operands
  if (!heun && energy->kind == EBM_ENERGY_GMM && gmm_mfma_supported(dim, energy->n_comp) &&
      !(dim == 32 && energy->n_comp <= 8)) {
    static const bool force_rows = ab_switch("EBM_GMM_ROWS");
    if (!force_rows)
      return launch_langevin_chain_gmm_mfma(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table,
                                            clamp_on, cmin, cmax, thin, traj, noise, seed, offset, (hipStream_t)stream);
  }
  return launch_langevin_chain_rows(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef,
                                    coef_table, clamp_on, cmin, cmax, thin, traj, noise, seed, offset,
                                    heun, nullptr, (hipStream_t)stream);
}

int ebm_langevin_chain_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                           int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                           const float* coef_table, int32_t clamp_on, float cmin, float cmax,
                           int32_t thin, float* traj, float* diag_partials, const float* noise, uint64_t seed,
                           uint64_t offset, void* stream) {
  return langevin_chain_impl("ebm_langevin_chain_f32", 0, energy, x, n_chains, dim, k_steps, eta, sqrt_eta,
                             noise_coef, coef_table, clamp_on, cmin, cmax, thin, traj, diag_partials, noise, seed, offset, stream);
}

int ebm_langevin_chain_dev_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                               float eta, float sqrt_eta, float noise_coef, const float* coef_table, int32_t clamp_on,
                               float cmin, float cmax, int32_t thin, float* traj, const uint64_t* rng_state,
                               uint64_t step_delta, void* stream) {
  const char* who = "ebm_langevin_chain_dev_f32";
  if (int r = check_energy(energy, dim, who)) return r;
  if (int r = check_state(x, n_chains, dim, who)) return r;
  if (k_steps < 0 || thin < 1) return fail(EBM_EINVAL, "%s: k_steps=%d thin=%d", who, k_steps, thin);
  if (!rng_state) return fail(EBM_EINVAL, "%s: rng_state is NULL", who);
  if (energy->kind != EBM_ENERGY_MLP)
    return fail(EBM_EKIND, "%s: device-resident RNG coordinates are taken by the EBM_ENERGY_MLP chain kernels only", who);
  if (n_chains == 0 || k_steps == 0) return 0;
  if ((coef_table && !aligned16(coef_table)) || (traj && !aligned16(traj)))
    return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  return launch_langevin_chain_mlp(*energy, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin, cmax,
                                   thin, traj, nullptr, 0, step_delta, nullptr, (hipStream_t)stream, rng_state);
}

int ebm_langevin_heun_chain_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                                int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                                const float* coef_table, int32_t clamp_on, float cmin, float cmax,
                                int32_t thin, float* traj, float* diag_partials, const float* noise, uint64_t seed,
                                uint64_t offset, void* stream) {
  return langevin_chain_impl("ebm_langevin_heun_chain_f32", 1, energy, x, n_chains, dim, k_steps, eta, sqrt_eta,
                             noise_coef, coef_table, clamp_on, cmin, cmax, thin, traj, diag_partials, noise, seed, offset, stream);
}

int ebm_hmc_chain_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                      int32_t n_mh, int32_t n_leapfrog, float eps, const float* eps_table,
                      int32_t mass_kind, double mass_scalar, const float* mass_diag, int32_t thin,
                      float* traj, float* diag_partials, uint8_t* accept_mask, uint32_t* accept_count,
                      const float* p_noise, const float* u, uint64_t seed, uint64_t offset,
                      void* stream) {
  const char* who = "ebm_hmc_chain_f32";
  if (int r = check_energy(energy, dim, who)) return r;
  if (int r = check_state(x, n_chains, dim, who)) return r;
  if (n_mh < 0 || thin < 1 || n_leapfrog < 1)
    return fail(EBM_EINVAL, "%s: n_mh=%d thin=%d n_leapfrog=%d", who, n_mh, thin, n_leapfrog);
  if (mass_kind < EBM_MASS_NONE || mass_kind > EBM_MASS_DIAG || (mass_kind == EBM_MASS_DIAG && !mass_diag))
    return fail(EBM_EINVAL, "%s: bad mass specification (kind %d)", who, mass_kind);
  if ((p_noise == nullptr) != (u == nullptr))
    return fail(EBM_EINVAL, "%s: p_noise and u must be given together", who);
  if (n_chains == 0 || n_mh == 0) return 0;
  if ((traj && !aligned16(traj)) || (p_noise && !aligned16(p_noise)) || (diag_partials && !aligned16(diag_partials)))
    return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  if (n_mh / thin == 0) diag_partials = nullptr;
  if (diag_partials) {
    diag::DiagArgs d;
    const DiagFamily fam = plan_diag(*energy, EBM_DIAG_HMC, n_chains, dim, p_noise != nullptr, traj != nullptr, d);
    if (fam != kDiagHmcRows && fam != kDiagMlp)
      return fail(energy->kind == EBM_ENERGY_MLP ? EBM_EKIND : EBM_EDIM,
                  "%s: no in-kernel diagnostics for this energy / dim %d (see ebm_diag_layout)", who, dim);
  }
  if (energy->kind == EBM_ENERGY_MLP)
    return launch_hmc_chain_mlp(*energy, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar,
                                mass_diag, thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, diag_partials,
                                (hipStream_t)stream);
  return launch_hmc_chain(*energy, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind,
                          mass_scalar, mass_diag, thin, traj, accept_mask, accept_count, p_noise, u,
                          seed, offset, diag_partials, (hipStream_t)stream);
}

int ebm_hmc_chain_audit_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog,
                            float eps, const float* eps_table, int32_t mass_kind, double mass_scalar, const float* mass_diag,
                            int32_t thin, float* traj, uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise,
                            const float* u, uint64_t seed, uint64_t offset, void* stream) {
  const char* who = "ebm_hmc_chain_audit_f32";
  if (int r = check_energy(energy, dim, who)) return r;
  if (int r = check_state(x, n_chains, dim, who)) return r;
  if (n_mh < 0 || thin < 1 || n_leapfrog < 1)
    return fail(EBM_EINVAL, "%s: n_mh=%d thin=%d n_leapfrog=%d", who, n_mh, thin, n_leapfrog);
  if (mass_kind < EBM_MASS_NONE || mass_kind > EBM_MASS_DIAG || (mass_kind == EBM_MASS_DIAG && !mass_diag))
    return fail(EBM_EINVAL, "%s: bad mass specification (kind %d)", who, mass_kind);
  if ((p_noise == nullptr) != (u == nullptr)) return fail(EBM_EINVAL, "%s: p_noise and u must be given together", who);
  if (n_chains == 0 || n_mh == 0) return 0;
  if ((traj && !aligned16(traj)) || (p_noise && !aligned16(p_noise))) return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  return launch_hmc_chain_audit(*energy, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag, thin, traj,
                                accept_mask, accept_count, p_noise, u, seed, offset, (hipStream_t)stream);
}

int ebm_leapfrog_kick_drift_f32(const float* x, const float* p, const float* force, float* x_new,
                                float* p_half, int64_t n_chains, int32_t dim, float eps,
                                int32_t mass_kind, double mass_scalar, const float* mass_diag,
                                int32_t safe, void* stream) {
  const char* who = "ebm_leapfrog_kick_drift_f32";
  if (n_chains < 0 || dim < 1) return fail(EBM_EINVAL, "%s: bad shape", who);
  if (n_chains == 0) return 0;
  if (!x || !p || !force || !x_new || !p_half) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  if (mass_kind == EBM_MASS_DIAG && !mass_diag) return fail(EBM_EINVAL, "%s: diagonal mass is NULL", who);
  if (!aligned16(x) || !aligned16(p) || !aligned16(force) || !aligned16(x_new) || !aligned16(p_half))
    return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  return launch_leapfrog_kick_drift(x, p, force, x_new, p_half, n_chains, dim, eps, mass_kind,
                                    mass_scalar, mass_diag, safe, (hipStream_t)stream);
}

int ebm_leapfrog_kick_f32(float* x_new, const float* p_half, const float* force, float* p_new,
                          int64_t n_elem, float eps, int32_t safe, void* stream) {
  const char* who = "ebm_leapfrog_kick_f32";
  if (n_elem < 0) return fail(EBM_EINVAL, "%s: n_elem < 0", who);
  if (n_elem == 0) return 0;
  if (!x_new || !p_half || !force || !p_new) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  if (!aligned16(x_new) || !aligned16(p_half) || !aligned16(force) || !aligned16(p_new))
    return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  return launch_leapfrog_kick(x_new, p_half, force, p_new, n_elem, eps, safe, (hipStream_t)stream);
}

int ebm_hmc_accept_f32(float* x, const float* x_prop, const float* h0, const float* h1,
                       const float* u, uint8_t* accept_mask, uint32_t* accept_count,
                       int64_t n_chains, int32_t dim, uint64_t seed, uint64_t offset, void* stream) {
  const char* who = "ebm_hmc_accept_f32";
  if (n_chains < 0 || dim < 1) return fail(EBM_EINVAL, "%s: bad shape", who);
  if (n_chains == 0) return 0;
  if (!x || !x_prop || !h0 || !h1) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  return launch_hmc_accept(x, x_prop, h0, h1, u, accept_mask, accept_count, n_chains, dim, seed,
                           offset, nullptr, (hipStream_t)stream);
}

int ebm_hmc_accept_dev_f32(float* x, const float* x_prop, const float* h0, const float* h1,
                           uint8_t* accept_mask, uint32_t* accept_count, int64_t n_chains, int32_t dim,
                           const uint64_t* rng_state, uint64_t step_delta, void* stream) {
  const char* who = "ebm_hmc_accept_dev_f32";
  if (n_chains < 0 || dim < 1) return fail(EBM_EINVAL, "%s: bad shape", who);
  if (n_chains == 0) return 0;
  if (!x || !x_prop || !h0 || !h1 || !rng_state) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  return launch_hmc_accept(x, x_prop, h0, h1, nullptr, accept_mask, accept_count, n_chains, dim, 0,
                           step_delta, rng_state, (hipStream_t)stream);
}

int ebm_descent_chain_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                          int32_t k_steps, float eta, const float* eta_table, int32_t nesterov,
                          float momentum, int32_t thin, float* traj, void* stream) {
  const char* who = "ebm_descent_chain_f32";
  if (int r = check_energy(energy, dim, who)) return r;
  if (int r = reject_mlp(energy, who)) return r;
  if (int r = check_state(x, n_chains, dim, who)) return r;
  if (k_steps < 0 || thin < 1) return fail(EBM_EINVAL, "%s: k_steps=%d thin=%d", who, k_steps, thin);
  if (n_chains == 0 || k_steps == 0) return 0;
  if (traj && !aligned16(traj)) return fail(EBM_EINVAL, "%s: traj must be 16-byte aligned", who);
  return launch_descent_chain(*energy, x, n_chains, dim, k_steps, eta, eta_table, nesterov, momentum, thin,
                              traj, (hipStream_t)stream);
}

int ebm_descent_step_f32(const float* x, const float* grad, float* v, float* out, int64_t n_elem,
                         float eta, float momentum, void* stream) {
  const char* who = "ebm_descent_step_f32";
  if (n_elem < 0) return fail(EBM_EINVAL, "%s: n_elem < 0", who);
  if (n_elem == 0) return 0;
  if (!x || !grad || !out) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  if (!aligned16(x) || !aligned16(grad) || !aligned16(out) || (v && !aligned16(v)))
    return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  return launch_descent_step(x, grad, v, out, n_elem, eta, momentum, (hipStream_t)stream);
}

int ebm_lookahead_f32(const float* x, const float* v, float* out, int64_t n_elem, float momentum,
                      void* stream) {
  const char* who = "ebm_lookahead_f32";
  if (n_elem < 0) return fail(EBM_EINVAL, "%s: n_elem < 0", who);
  if (n_elem == 0) return 0;
  if (!x || !v || !out) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  if (!aligned16(x) || !aligned16(v) || !aligned16(out)) return fail(EBM_EINVAL, "%s: pointers must be 16-byte aligned", who);
  return launch_lookahead(x, v, out, n_elem, momentum, (hipStream_t)stream);
}

int ebm_pcd_gather_f32(const float* buffer, int64_t buffer_size, int32_t dim, float* out, int64_t batch,
                       int64_t stride, const int64_t* offsets, int64_t* rows_out, uint64_t seed,
                       uint64_t offset, void* stream) {
  const char* who = "ebm_pcd_gather_f32";
  if (buffer_size < 1 || dim < 1 || batch < 0 || stride < 1 || stride > 0x7fffffffLL)
    return fail(EBM_EINVAL, "%s: bad sizes (buffer %lld, dim %d, batch %lld, stride %lld)", who, (long long)buffer_size, dim,
                (long long)batch, (long long)stride);
  if (batch == 0) return 0;
  if (!buffer || !out) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  return launch_pcd_gather(buffer, buffer_size, dim, out, batch, stride, offsets, rows_out, seed, offset, nullptr,
                           (hipStream_t)stream);
}

int ebm_pcd_gather_dev_f32(const float* buffer, int64_t buffer_size, int32_t dim, float* out, int64_t batch,
                           int64_t stride, int64_t* rows_out, const uint64_t* rng_state, uint64_t step_delta,
                           void* stream) {
  const char* who = "ebm_pcd_gather_dev_f32";
  if (buffer_size < 1 || dim < 1 || batch < 0 || stride < 1 || stride > 0x7fffffffLL)
    return fail(EBM_EINVAL, "%s: bad sizes (buffer %lld, dim %d, batch %lld, stride %lld)", who, (long long)buffer_size, dim,
                (long long)batch, (long long)stride);
  if (batch == 0) return 0;
  if (!buffer || !out || !rng_state) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  return launch_pcd_gather(buffer, buffer_size, dim, out, batch, stride, nullptr, rows_out, 0, step_delta, rng_state,
                           (hipStream_t)stream);
}

int64_t ebm_cd_loss_work_bytes(void) { return cd_loss_work_bytes(); }

int ebm_cd_loss_f32(const float* e_both, int64_t n, float reg, void* work, float* loss_out, float* finite_out, void* stream) {
  const char* who = "ebm_cd_loss_f32";
  if (n < 1) return fail(EBM_EINVAL, "%s: n < 1", who);
  if (!e_both || !work || !loss_out || !finite_out || (reinterpret_cast<uintptr_t>(work) & 7) != 0)
    return fail(EBM_EINVAL, "%s: NULL pointer, or work not 8-byte aligned", who);
  return launch_cd_loss(e_both, n, reg, work, loss_out, finite_out, (hipStream_t)stream);
}

int ebm_cd_loss_backward_f32(const float* e_both, int64_t n, float reg, const float* upstream, const float* finite, float* seed_out,
                             void* stream) {
  const char* who = "ebm_cd_loss_backward_f32";
  if (n < 1) return fail(EBM_EINVAL, "%s: n < 1", who);
  if (!e_both || !upstream || !finite || !seed_out) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  return launch_cd_loss_seed(e_both, n, reg, upstream, finite, seed_out, (hipStream_t)stream);
}

int ebm_pcd_start_points_f32(const float* buffer, int64_t buffer_size, int32_t dim, float* out, int64_t batch, int64_t stride,
                             int64_t n_noise, float noise_scale, uint64_t seed, uint64_t step, const uint64_t* rng_state, void* stream) {
  const char* who = "ebm_pcd_start_points_f32";
  if (buffer_size < 1 || dim < 1 || batch < 0 || batch > 0x7fffffffLL || stride < 1 || stride > 0x7fffffffLL || n_noise < 0 || n_noise > batch)
    return fail(EBM_EINVAL, "%s: bad sizes (buffer %lld, dim %d, batch %lld, stride %lld, n_noise %lld)", who, (long long)buffer_size, dim,
                (long long)batch, (long long)stride, (long long)n_noise);
  if (batch == 0) return 0;
  if (!buffer || !out) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  return launch_pcd_start_points(buffer, buffer_size, dim, out, batch, stride, n_noise, noise_scale, seed, step, rng_state, (hipStream_t)stream);
}

int ebm_gmm_active_columns_i32(const float* means, int32_t n_comp, int32_t dim, int32_t* out, void* stream) {
  const char* who = "ebm_gmm_active_columns_i32";
  if (n_comp < 1 || dim < 4 || dim > 32 || (dim % 4) != 0)
    return fail(EBM_EINVAL, "%s: bad sizes (n_comp %d, dim %d: a multiple of 4 up to 32)", who, n_comp, dim);
  if (!means || !out) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  return launch_gmm_active_columns(means, n_comp, dim, out, (hipStream_t)stream);
}

int ebm_pcd_scatter_f32(float* buffer, int64_t buffer_size, int32_t dim, const float* samples, int64_t batch,
                        int64_t write_pos, void* stream) {
  const char* who = "ebm_pcd_scatter_f32";
  if (buffer_size < 1 || dim < 1 || batch < 0 || batch > buffer_size || write_pos < 0 || write_pos >= buffer_size)
    return fail(EBM_EINVAL, "%s: bad sizes (buffer %lld, batch %lld, write_pos %lld)", who, (long long)buffer_size,
                (long long)batch, (long long)write_pos);
  if (batch == 0) return 0;
  if (!buffer || !samples) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  return launch_pcd_scatter(buffer, buffer_size, dim, samples, batch, write_pos, nullptr, (hipStream_t)stream);
}

int ebm_pcd_scatter_dev_f32(float* buffer, int64_t buffer_size, int32_t dim, const float* samples, int64_t batch,
                            const int64_t* write_pos, void* stream) {
  const char* who = "ebm_pcd_scatter_dev_f32";
  if (buffer_size < 1 || dim < 1 || batch < 0 || batch > buffer_size)
    return fail(EBM_EINVAL, "%s: bad sizes (buffer %lld, batch %lld)", who, (long long)buffer_size, (long long)batch);
  if (batch == 0) return 0;
  if (!buffer || !samples || !write_pos) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  return launch_pcd_scatter(buffer, buffer_size, dim, samples, batch, 0, write_pos, (hipStream_t)stream);
}

int ebm_energy_grad_f32(const ebm_energy_t* energy, const float* x, int64_t n_chains, int32_t dim,
                        float* energy_out, float* grad_out, void* stream) {
  const char* who = "ebm_energy_grad_f32";
  if (int r = check_energy(energy, dim, who)) return r;
  if (int r = check_state(x, n_chains, dim, who)) return r;
  if (n_chains == 0) return 0;
  if (grad_out && !aligned16(grad_out)) return fail(EBM_EINVAL, "%s: grad_out must be 16-byte aligned", who);
  if (energy->kind == EBM_ENERGY_MLP)
    return launch_energy_grad_mlp(*energy, x, n_chains, dim, energy_out, grad_out, (hipStream_t)stream);
  if (energy->kind == EBM_ENERGY_GAUSSIAN && gauss_big_supported(dim)) {  // above 128 dims: one contraction pass on the matrix cores
    static const bool force_rows = ab_switch("EBM_GAUSS_ROWS");
    if (!force_rows) return launch_energy_grad_gauss_big(*energy, x, n_chains, dim, energy_out, grad_out, (hipStream_t)stream);
  }
  return launch_energy_grad(*energy, x, n_chains, dim, energy_out, grad_out, (hipStream_t)stream);
}

int ebm_mlp_backward_acts_f32(const ebm_energy_t* energy, const float* x, int64_t n_chains, int32_t dim, const float* seed,
                              float* energy_out, float* grad_out, float* acts, void* stream) {
  const char* who = "ebm_mlp_backward_acts_f32";
  if (int r = check_energy(energy, dim, who)) return r;
  if (int r = check_state(x, n_chains, dim, who)) return r;
  if (energy->kind != EBM_ENERGY_MLP) return fail(EBM_EKIND, "%s: EBM_ENERGY_MLP only", who);
  if (!energy->dev0) return fail(EBM_EINVAL, "%s: packed MLP parameters pointer is NULL", who);
  if ((energy->n_comp != 64 && energy->n_comp != 128) || dim < 1 || dim > 64)
    return fail(EBM_EDIM, "%s: hidden width 64 or 128 and dim <= 64 (got %d, %d)", who, energy->n_comp, dim);
  if (n_chains == 0) return 0;
  if (!acts || !aligned16(acts) || (grad_out && !aligned16(grad_out))) return fail(EBM_EINVAL, "%s: acts / grad_out must be 16-byte aligned pointers", who);
  if (((n_chains + 127) / 128 * 128) * 20 >= (1LL << 32)) return fail(EBM_EINVAL, "%s: too many rows for 32-bit lane offsets", who);  // (a bound kept from the [4][H][n] layout)
  return launch_mlp_backward_acts(energy->n_comp, energy->dev0, x, n_chains, dim, seed, energy_out, grad_out, acts, (hipStream_t)stream, who);
}

int64_t ebm_mlp_param_grads_work_f32(int32_t hidden, int32_t dim, int64_t n_rows) {
  if ((hidden != 64 && hidden != 128) || dim < 1 || dim > 64 || n_rows < 0) return 0;
  return mlp_param_grads_work_floats(hidden, dim, n_rows);
}

int ebm_mlp_param_grads_f32(const float* acts, int64_t n_rows, int32_t hidden, const float* x, int32_t dim, const float* seed,
                            const float* w3, float* work, int64_t work_floats, float* grads_out, void* stream) {
  const char* who = "ebm_mlp_param_grads_f32";
  if ((hidden != 64 && hidden != 128) || dim < 1 || dim > 64)
    return fail(EBM_EDIM, "%s: hidden width 64 or 128 and dim <= 64 (got %d, %d)", who, hidden, dim);
  if (n_rows < 1) return fail(EBM_EINVAL, "%s: no rows", who);
  if (!acts || !aligned16(acts) || !x || !grads_out || !work || !w3) return fail(EBM_EINVAL, "%s: NULL pointer, or acts not 16-byte aligned", who);
  if (work_floats < mlp_param_grads_work_floats(hidden, dim, n_rows))
    return fail(EBM_EINVAL, "%s: workspace of %lld floats, need ebm_mlp_param_grads_work_f32() = %lld", who, (long long)work_floats,
                (long long)mlp_param_grads_work_floats(hidden, dim, n_rows));
  return launch_mlp_param_grads(hidden, acts, x, n_rows, dim, seed, w3, work, grads_out, (hipStream_t)stream, who);
}

int ebm_chain_stats_f32(const float* x, int64_t n_chains, int32_t dim, float* mean_out,
                        float* var_out, double* work, void* stream) {
  const char* who = "ebm_chain_stats_f32";
  if (int r = check_state(x, n_chains, dim, who)) return r;
  if (!mean_out || !var_out || !work) return fail(EBM_EINVAL, "%s: NULL output/workspace", who);
  if (n_chains == 0) return fail(EBM_EINVAL, "%s: no chains", who);
  return launch_chain_stats(x, n_chains, dim, mean_out, var_out, work, (hipStream_t)stream);
}

int ebm_noise_fill_f32(float* out, int64_t n_elem, int32_t kind, uint64_t seed, uint64_t offset,
                       void* stream) {
  const char* who = "ebm_noise_fill_f32";
  if (n_elem < 0) return fail(EBM_EINVAL, "%s: n_elem < 0", who);
  if (n_elem == 0) return 0;
  if (!out || !aligned16(out)) return fail(EBM_EINVAL, "%s: out must be a 16-byte aligned pointer", who);
  if (kind < EBM_NOISE_NORMAL || kind > EBM_NOISE_RAW_U32) return fail(EBM_EINVAL, "%s: bad kind %d", who, kind);
  return launch_noise_fill(out, n_elem, kind, seed, offset, nullptr, (hipStream_t)stream);
}

int ebm_noise_fill_dev_f32(float* out, int64_t n_elem, int32_t kind, const uint64_t* rng_state,
                           uint64_t step_delta, void* stream) {
  const char* who = "ebm_noise_fill_dev_f32";
  if (n_elem < 0) return fail(EBM_EINVAL, "%s: n_elem < 0", who);
  if (n_elem == 0) return 0;
  if (!out || !aligned16(out)) return fail(EBM_EINVAL, "%s: out must be a 16-byte aligned pointer", who);
  if (!rng_state) return fail(EBM_EINVAL, "%s: rng_state is NULL", who);
  if (kind < EBM_NOISE_NORMAL || kind > EBM_NOISE_RAW_U32) return fail(EBM_EINVAL, "%s: bad kind %d", who, kind);
  return launch_noise_fill(out, n_elem, kind, 0, step_delta, rng_state, (hipStream_t)stream);
}

int ebm_diag_layout(const ebm_energy_t* energy, int32_t sampler, int64_t n_chains, int32_t dim, int32_t injected_noise,
                    int32_t with_traj, int64_t* n_blocks, int32_t* slots, int32_t* block_elems) {
  const char* who = "ebm_diag_layout";
  if (int r = check_energy(energy, dim, who)) return r;
  if (sampler < EBM_DIAG_LANGEVIN || sampler > EBM_DIAG_HMC) return fail(EBM_EINVAL, "%s: unknown sampler kind %d", who, sampler);
  if (n_chains < 1 || dim < 1 || !n_blocks || !slots || !block_elems)
    return fail(EBM_EINVAL, "%s: bad shape [%lld, %d] or NULL output", who, (long long)n_chains, dim);
  diag::DiagArgs d;
  if (plan_diag(*energy, sampler, n_chains, dim, injected_noise != 0, with_traj != 0, d) == kDiagNone)
    return fail(energy->kind == EBM_ENERGY_MLP ? EBM_EKIND : EBM_EDIM, "%s: no in-kernel diagnostics for this energy / dim %d", who,
                dim);
  *n_blocks = d.n_blocks;
  *slots = d.S;
  *block_elems = d.E;
  return 0;
}

int ebm_diag_finish_f32(const float* diag_partials, int32_t n_kept, int64_t n_blocks, int32_t slots, int32_t block_elems,
                        int64_t n_chains, int32_t dim, float* mean_out, float* var_out, float* energy_out, float* accept_out,
                        double* work, void* stream) {
  const char* who = "ebm_diag_finish_f32";
  if (n_kept < 0) return fail(EBM_EINVAL, "%s: n_kept < 0", who);
  if (n_kept == 0) return 0;
  if (!diag_partials || !mean_out || !var_out || !work) return fail(EBM_EINVAL, "%s: NULL records / outputs / workspace", who);
  if (block_elems < 0) {  // records of interleaved classes (shifted rows): -32 dim, 4 / gcd(dim, 4) classes
    const int K = diag::diag_classes(dim);
    if (n_chains < 1 || dim < 1 || K < 2 || block_elems != -32 * dim || slots != dim || n_blocks != ceil_div64(n_chains, 32 * (int64_t)K) * K)
      return fail(EBM_EINVAL, "%s: inconsistent interleaved layout (n_blocks %lld, slots %d, block_elems %d for [%lld, %d])", who,
                  (long long)n_blocks, slots, block_elems, (long long)n_chains, dim);
    return launch_diag_finish(diag_partials, n_kept, n_blocks, slots, block_elems, n_chains, dim, mean_out, var_out, energy_out,
                              accept_out, work, (hipStream_t)stream);
  }
  if (n_chains < 1 || dim < 1 || n_blocks < 1 || slots < 1 || block_elems < 1 || slots != (dim < block_elems ? dim : block_elems) ||
      (block_elems % dim != 0 && dim % block_elems != 0) || n_blocks != ceil_div64(n_chains * (int64_t)dim, block_elems))
    return fail(EBM_EINVAL, "%s: inconsistent layout (n_blocks %lld, slots %d, block_elems %d for [%lld, %d])", who,
                (long long)n_blocks, slots, block_elems, (long long)n_chains, dim);
  return launch_diag_finish(diag_partials, n_kept, n_blocks, slots, block_elems, n_chains, dim, mean_out, var_out, energy_out,
                            accept_out, work, (hipStream_t)stream);
}

int ebm_probe_valu_f32(float* out, int32_t blocks, int32_t iters, void* stream) {
  const char* who = "ebm_probe_valu_f32";
  if (!out || blocks < 1 || iters < 1) return fail(EBM_EINVAL, "%s: out is NULL or blocks/iters < 1", who);
  return launch_probe_valu(out, blocks, iters, (hipStream_t)stream);
}

int ebm_probe_issue_f32(float* out, int32_t blocks, int32_t iters, int32_t kind, void* stream) {
  const char* who = "ebm_probe_issue_f32";
  if (!out || blocks < 1 || iters < 1) return fail(EBM_EINVAL, "%s: out is NULL or blocks/iters < 1", who);
  if (kind < 0 || kind > 7) return fail(EBM_EINVAL, "%s: kind %d (0 fma | 1 mad_u64_u32 | 2 transcendental | 3 pk_fma | 4 bitop3 | 5 pk_fma, VGPR operands | 6 pk_mul | 7 the lean loop's mix)", who, kind);
  return launch_probe_issue(out, blocks, iters, kind, (hipStream_t)stream);
}

size_t ebm_mlp_w1_image_bytes(int32_t hidden, int32_t dim) { return mlp_w1_image_bytes(hidden, dim); }

size_t ebm_gauss_prec_image_bytes(int32_t dim) { return gauss_prec_image_bytes(dim); }

int ebm_gauss_prec_image_f32(const float* prec, int32_t dim, void* image, void* stream) {
  return launch_gauss_prec_image(prec, dim, image, (hipStream_t)stream, "ebm_gauss_prec_image_f32");
}

int ebm_mlp_w1_image_f32(const float* params, int32_t hidden, int32_t dim, void* image, void* stream) {
  return launch_mlp_w1_image(params, hidden, dim, image, (hipStream_t)stream, "ebm_mlp_w1_image_f32");
}

}  // extern "C"
