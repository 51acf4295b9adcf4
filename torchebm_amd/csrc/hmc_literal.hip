// The audit instantiation of the HMC transition kernel for the element-wise energies (hmc_kernel.h: leapfrog_literal) --
// ebm_hmc_chain_audit_f32, for tests: the reference's leapfrog sequence with no fused multiply-add, no merged kick and the drift
// divided by max(m, 1e-10) on every step.  Reference: torchebm/integrators/leapfrog.py:156-185, samplers/hmc.py:243-312.
#include "hmc_kernel.h"

namespace ebm {
namespace hmc {
void launch_literal_double_well(const rows::Geometry& geo, dim3 grid, size_t smem, hipStream_t st, const HmcArgs& a) {
  launch_literal<EBM_ENERGY_DOUBLE_WELL>(geo, grid, smem, st, a);
}
void launch_literal_harmonic(const rows::Geometry& geo, dim3 grid, size_t smem, hipStream_t st, const HmcArgs& a) {
  launch_literal<EBM_ENERGY_HARMONIC>(geo, grid, smem, st, a);
}
}  // namespace hmc
}  // namespace ebm
