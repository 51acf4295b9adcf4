// HMC kernels for one energy (see hmc_kernel.h); split out so the energies build in parallel.
#include "hmc_kernel.h"

namespace ebm {
namespace hmc {
void launch_double_well(const rows::Geometry& geo, dim3 grid, size_t smem, hipStream_t st, const HmcArgs& a) {
  launch_kind<EBM_ENERGY_DOUBLE_WELL, false>(geo, grid, smem, st, a);
}
}  // namespace hmc
}  // namespace ebm
