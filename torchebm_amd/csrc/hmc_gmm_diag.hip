// HMC kernels for one energy WITH the in-kernel diagnostics records (hmc_kernel.h: DIAG; diag.h); their own
// translation unit so that they build beside the plain ones.
#include "hmc_kernel.h"

namespace ebm {
namespace hmc {
void launch_gmm_diag(const rows::Geometry& geo, dim3 grid, size_t smem, hipStream_t st, const HmcArgs& a) {
  launch_kind<EBM_ENERGY_GMM, true>(geo, grid, smem, st, a);
}
}  // namespace hmc
}  // namespace ebm
