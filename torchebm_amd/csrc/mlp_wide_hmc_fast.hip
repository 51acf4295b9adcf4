// FAST instantiations of the wide-MLP HMC transition kernel (MODE 2 shapes, no / scalar mass: the plain call) -- see
// mlp_wide_hmc_body.h.
#include "mlp_wide_hmc_body.h"

namespace ebm {
namespace widemlp {

#define EBM_HMC_FAST(HTV, DTV)                                                                       \
  template <>                                                                                        \
  int launch_hmc_fast<HTV, DTV>(const WideHmcArgs& a, hipStream_t st, const char* who) {             \
    return launch_hmc_variant<HTV, DTV, false, true>(a, st, who);                                    \
  }
EBM_HMC_FAST(2, 1) EBM_HMC_FAST(2, 2) EBM_HMC_FAST(2, 3) EBM_HMC_FAST(2, 4)
EBM_HMC_FAST(4, 1) EBM_HMC_FAST(4, 2)
#undef EBM_HMC_FAST

}  // namespace widemlp
}  // namespace ebm
