// Dense Gaussians above 128 dims: the tiled kernels' instantiations, the dispatch and the entry points (gauss_big_body.h has the
// kernels and their description; the register-resident kernels are instantiated in gauss_res.hip: two translation units compile in
// parallel, 2m13 -> ~1m10 each).
#include "gauss_big_body.h"

namespace ebm {
using namespace gbig;
int launch_gauss_res(int tiles, const gbig::BigArgs& a, hipStream_t st);  // gauss_res.hip
namespace {
int dispatch_big(const BigArgs& a, int32_t dim, hipStream_t st) {
  const int tiles = (dim + 31) / 32;  // 5 .. 16
#ifndef EBM_BIG_TILED_ONLY
  if (a.k_steps > 0) {
    // up to seven tiles the register-resident kernel, eight tiles the tiled one (same box, 2^17 chains x 20 steps, ms:
    // dims 132 / 160 / 192 / 224 / 256: 1.37 / 1.42 / 1.89 / 2.55 / 3.48 resident, 1.48 / 1.67 / 2.23 / 2.51 / 3.02 tiled) --
    // unless the pre-split image is there (round 4): the resident kernel then has no slab work, and eight tiles (dim 256) stay in
    // registers too: 2.84 (tiled, image) -> 2.37 ms, no state traffic in the step loop (with or without records: the same chains)
    if (tiles <= 7) return launch_gauss_res(tiles, a, st);  // gauss_res.hip
    if (tiles == 8 && a.prec_image && (reinterpret_cast<uintptr_t>(a.prec_image) & 15) == 0)
      return launch_gauss_res(tiles, a, st);
  }
#endif
  if (tiles <= 8) {
    switch (tiles) {
      case 5: return launch_big<5, 1>(a, st);
      case 6: return launch_big<6, 1>(a, st);
      case 7: return launch_big<7, 1>(a, st);
      default: return launch_big<8, 1>(a, st);
    }
  }
  switch ((tiles + 1) / 2) {  // two slices of ceil(tiles / 2) out tiles (the second one may end in a zero tile)
    case 5: return launch_big<5, 2>(a, st);
    case 6: return launch_big<6, 2>(a, st);
    case 7: return launch_big<7, 2>(a, st);
    default: return launch_big<8, 2>(a, st);
  }
}

}  // namespace

bool gauss_big_supported(int32_t dim) { return dim > 128 && dim <= 512 && (dim % 4) == 0; }
// records: one per wave-tile of 32 chains and kept step, as on the other matrix-layout kernels
bool gauss_big_diag_plan(int64_t n_chains, int32_t dim, diag::DiagArgs& d) {
  return gauss_big_supported(dim) && diag::plan(n_chains, dim, 32 * (int64_t)dim, d);
}

int launch_langevin_chain_gauss_big(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                                    float eta, float sqrt_eta, float noise_coef, const float* coef_table,
                                    int clamp_on, float cmin, float cmax, int32_t thin, float* traj,
                                    const float* noise, uint64_t seed, uint64_t offset, float* diag_partials, hipStream_t st) {
  if (!gauss_big_supported(dim)) return fail(EBM_EDIM, "ebm_langevin_chain_f32: the tiled Gaussian kernel takes dims 132 .. 512 in steps of 4, not %d", dim);
  BigArgs a{};
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.noise = noise; a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj;
  a.mean = e.dev0; a.prec = e.dev1;
  a.prec_image = reinterpret_cast<const char*>(e.aux);  // (the resident kernels up to seven tiles do not read it)
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  a.energy_out = nullptr; a.grad_out = nullptr;
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0};
  if (diag_partials) {
    if (!gauss_big_diag_plan(n_chains, dim, a.diag)) return fail(EBM_EDIM, "ebm_langevin_chain_f32: no records layout for dim %d", dim);
    a.diag.partials = diag_partials;
  }
  return dispatch_big(a, dim, st);
}

// E(x) and / or its gradient for the same widths (ebm_energy_grad_f32): one contraction pass of the tiled kernel, nothing updated
int launch_energy_grad_gauss_big(const ebm_energy_t& e, const float* x, int64_t n_chains, int32_t dim, float* energy_out, float* grad_out,
                                 hipStream_t st) {
  if (!gauss_big_supported(dim)) return fail(EBM_EDIM, "ebm_energy_grad_f32: the tiled Gaussian kernel takes dims 132 .. 512 in steps of 4, not %d", dim);
  BigArgs a{};
  a.x = const_cast<float*>(x); a.n_chains = n_chains; a.dim = dim; a.k_steps = 0;
  a.eta = 0.0f; a.sqrt_eta = 0.0f; a.noise_coef = 0.0f; a.table = nullptr; a.noise = nullptr;
  a.clamp_on = 0; a.cmin = 0.0f; a.cmax = 0.0f; a.thin = 1; a.n_kept = 0; a.traj = nullptr;
  a.mean = e.dev0; a.prec = e.dev1; a.key = RngKey{0u, 0u}; a.step0 = 0;
  a.prec_image = reinterpret_cast<const char*>(e.aux);
  a.energy_out = energy_out; a.grad_out = grad_out;
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0};
  return dispatch_big(a, dim, st);
}

}  // namespace ebm
