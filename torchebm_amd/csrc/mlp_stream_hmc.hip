// H = 256 instantiations of the wide-MLP HMC transition kernel (weights streamed from L2) -- see mlp_wide_hmc.hip.
#include "mlp_wide_hmc_body.h"

namespace ebm {

int launch_hmc_mlp_stream(const widemlp::WideHmcArgs& a, int dt, hipStream_t st, const char* who) {
  switch (dt) {
    case 1: return widemlp::launch_hmc_mass<8, 1>(a, st, who);
    case 2: return widemlp::launch_hmc_mass<8, 2>(a, st, who);
    case 3: return widemlp::launch_hmc_mass<8, 3>(a, st, who);
    default: return widemlp::launch_hmc_mass<8, 4>(a, st, who);
  }
}

}  // namespace ebm
