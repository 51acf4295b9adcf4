// MODE 3 instantiations of the wide-MLP HMC transition kernel (mlp_wide_hmc_body.h): hidden width 128 at input widths
// 65 .. 128 on the bf16 matrix pipe, the pre-split W1 image streamed slab by slab through LDS (mlp_b16.h "MODE 3",
// mlp_wide_slab.hip for the chain kernel and the image builder).
#include "mlp_wide_body.h"
#include "mlp_wide_hmc_body.h"

namespace ebm {
namespace widemlp {

#define EBM_HMC_SLAB(DTV, DM)                                                                      \
  template <>                                                                                      \
  int launch_hmc_slab<DTV, DM>(const WideHmcArgs& a, hipStream_t st, const char* who) {            \
    return launch_hmc_variant<4, DTV, DM, false, 3>(a, st, who);                                   \
  }
EBM_HMC_SLAB(3, false) EBM_HMC_SLAB(3, true) EBM_HMC_SLAB(4, false) EBM_HMC_SLAB(4, true)
#undef EBM_HMC_SLAB

}  // namespace widemlp
}  // namespace ebm
