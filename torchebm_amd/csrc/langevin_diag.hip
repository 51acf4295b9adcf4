// Element-wise Langevin chain kernels that also emit the per-block diagnostics records (diag.h) at the kept
// steps: the DIAG instantiations of the lean k-fused kernel, in their own translation unit so that they
// compile beside langevin.hip.  Reference: torchebm/samplers/langevin_dynamics.py:170-185.
#include "langevin_elem.h"

namespace ebm {

// Block geometry of the flat kernel: 256 lanes x one float4 = 1024 consecutive elements per workgroup.
bool elem_diag_supported(int32_t dim, bool has_noise, bool has_traj) {
  if (has_noise) return false;                       // the lean loop draws its own noise
  if (has_traj && (dim & 3) != 0) return false;      // float4 trajectory rows
  return (kBlock * 4) % dim == 0 || dim % (kBlock * 4) == 0;
}

bool elem_diag_plan(int64_t n_chains, int32_t dim, diag::DiagArgs& d) { return diag::plan(n_chains, dim, kBlock * 4, d); }

int launch_langevin_chain_elem_diag(int kind, float s0, float s1, float* x, int64_t n_chains, int32_t dim,
                                    int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                                    const float* coef_table, int clamp_on, float cmin, float cmax, int32_t thin,
                                    float* traj, uint64_t seed, uint64_t offset, int heun, float* diag_partials,
                                    hipStream_t st) {
  const char* who = heun ? "ebm_langevin_heun_chain_f32" : "ebm_langevin_chain_f32";
  ChainArgs a{};
  a.x = x; a.n_elem = n_chains * (int64_t)dim; a.dim = dim; a.k_steps = k_steps;
  a.c = StepCoef{eta, sqrt_eta, noise_coef};
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = nullptr;
  a.s0 = s0; a.s1 = s1;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  if (!elem_diag_plan(n_chains, dim, a.diag)) return fail(EBM_EDIM, "%s: diagnostics records need dim | 1024 or 1024 | dim on the flat kernel (dim %d)", who, dim);
  a.diag.partials = diag_partials;
  if (a.diag.n_blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "%s: state too large for one launch", who);
  const dim3 grid((unsigned)a.diag.n_blocks), block(kBlock);
  const int lds_generic = diag::lds_floats(a.diag.E, a.diag.S), lds_fast = 2 * diag::fast_lds_floats();
  const size_t smem = (size_t)(diag::fast_flat_ok(dim) ? lds_fast : lds_generic) * sizeof(float);
#define EBM_D_T(KIND, TB, CL, HE)                                                                                     \
  do {                                                                                                               \
    if (traj) hipLaunchKernelGGL((langevin_chain_lean_diag_kernel<KIND, TB, CL, true, HE>), grid, block, smem, st, a);   \
    else hipLaunchKernelGGL((langevin_chain_lean_diag_kernel<KIND, TB, CL, false, HE>), grid, block, smem, st, a);       \
  } while (0)
#define EBM_D_H(KIND, HE)                                       \
  do {                                                          \
    if (coef_table && clamp_on) EBM_D_T(KIND, true, true, HE);  \
    else if (coef_table) EBM_D_T(KIND, true, false, HE);        \
    else if (clamp_on) EBM_D_T(KIND, false, true, HE);          \
    else EBM_D_T(KIND, false, false, HE);                       \
  } while (0)
#define EBM_D(KIND)                    \
  do {                                 \
    if (heun) EBM_D_H(KIND, true);     \
    else EBM_D_H(KIND, false);         \
  } while (0)
  if (kind == EBM_ENERGY_DOUBLE_WELL) EBM_D(EBM_ENERGY_DOUBLE_WELL);
  else EBM_D(EBM_ENERGY_HARMONIC);
#undef EBM_D
#undef EBM_D_H
#undef EBM_D_T
  return check_launch(who);
}

}  // namespace ebm
