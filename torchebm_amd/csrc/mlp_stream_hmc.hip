// H = 256 instantiations of the wide-MLP HMC transition kernel (weights streamed from L2), scalar / identity mass -- see
// mlp_wide_hmc.hip; the diagonal-mass ones are in mlp_stream_hmc_diag.hip.
#include "mlp_wide_hmc_body.h"

namespace ebm {

int launch_hmc_mlp_stream_diag(const widemlp::WideHmcArgs& a, int dt, hipStream_t st, const char* who);  // mlp_stream_hmc_diag.hip

int launch_hmc_mlp_stream(const widemlp::WideHmcArgs& a, int dt, hipStream_t st, const char* who) {
  if (a.mass_kind == EBM_MASS_DIAG) return launch_hmc_mlp_stream_diag(a, dt, st, who);
  switch (dt) {
    case 1: return widemlp::launch_hmc_one<8, 1, false>(a, st, who);
    case 2: return widemlp::launch_hmc_one<8, 2, false>(a, st, who);
    case 3: return widemlp::launch_hmc_one<8, 3, false>(a, st, who);
    default: return widemlp::launch_hmc_one<8, 4, false>(a, st, who);
  }
}

}  // namespace ebm
