// Diagonal-mass instantiations of the wide-MLP HMC transition kernel for H = 64 / 128 -- see mlp_wide_hmc.hip.
#include "mlp_wide_hmc_body.h"

namespace ebm {

int launch_hmc_mlp_wide_diag(const widemlp::WideHmcArgs& a, int hidden, int dt, hipStream_t st, const char* who) {
#define EBM_WIDE_HMC_DIAG(HTV)                                              \
  switch (dt) {                                                             \
    case 1: return widemlp::launch_hmc_one<HTV, 1, true>(a, st, who);       \
    case 2: return widemlp::launch_hmc_one<HTV, 2, true>(a, st, who);       \
    case 3: return widemlp::launch_hmc_one<HTV, 3, true>(a, st, who);       \
    default: return widemlp::launch_hmc_one<HTV, 4, true>(a, st, who);      \
  }
  if (hidden == 64) { EBM_WIDE_HMC_DIAG(2) }
  EBM_WIDE_HMC_DIAG(4)
#undef EBM_WIDE_HMC_DIAG
}

}  // namespace ebm
