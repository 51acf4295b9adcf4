// FAST instantiations of the wide-MLP chain kernel WITH diagnostics records (FAST = 2; MODE 2 shapes) -- see mlp_wide_body.h.
#include "mlp_wide_body.h"

namespace ebm {
namespace widemlp {

#define EBM_FAST(HTV, DTV)                                                                        \
  template <>                                                                                     \
  int launch_fast_diag<HTV, DTV>(const WideArgs& a, hipStream_t st, const char* who) {            \
    return launch_variant<HTV, DTV, 2, 2>(a, st, who);                                            \
  }
EBM_FAST(2, 1) EBM_FAST(2, 2) EBM_FAST(2, 3) EBM_FAST(2, 4)
EBM_FAST(4, 1) EBM_FAST(4, 2)
#undef EBM_FAST

}  // namespace widemlp
}  // namespace ebm
