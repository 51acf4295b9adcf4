// H = 256, diagonal mass: instantiations of the wide-MLP HMC transition kernel -- see mlp_wide_hmc.hip.
#include "mlp_wide_hmc_body.h"

namespace ebm {

int launch_hmc_mlp_stream_diag(const widemlp::WideHmcArgs& a, int dt, hipStream_t st, const char* who) {
  switch (dt) {
    case 1: return widemlp::launch_hmc_one<8, 1, true>(a, st, who);
    case 2: return widemlp::launch_hmc_one<8, 2, true>(a, st, who);
    case 3: return widemlp::launch_hmc_one<8, 3, true>(a, st, who);
    default: return widemlp::launch_hmc_one<8, 4, true>(a, st, who);
  }
}

}  // namespace ebm
