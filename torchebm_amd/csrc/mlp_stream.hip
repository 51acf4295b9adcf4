// H = 256 instantiations of the wide-MLP chain kernel (STREAM: weights read from L2, s1 parked in LDS) -- see mlp_wide.hip.
#include "mlp_wide_body.h"

namespace ebm {

int launch_mlp_stream(const widemlp::WideArgs& a, int dt, hipStream_t st, const char* who) {
  switch (dt) {
    case 1: return widemlp::launch_one<8, 1>(a, st, who);
#ifndef EBM_STREAM_DT1_ONLY
    case 2: return widemlp::launch_one<8, 2>(a, st, who);
    case 3: return widemlp::launch_one<8, 3>(a, st, who);
#endif
    default: return widemlp::launch_one<8, 4>(a, st, who);
  }
}

}  // namespace ebm
