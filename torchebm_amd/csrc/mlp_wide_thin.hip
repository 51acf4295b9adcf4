// MODE 4 instantiations of the wide-MLP kernel (mlp_wide_body.h): input width <= 2 -- BASELINE config 5's shape, the 2-D energies of
// the reference's training examples (examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-31) --, hidden 64 / 128.
// W2's two contractions stay on the bf16 matrix pipe; the two with W1 (2 FMAs per hidden unit) run on the vector unit in exact fp32.
// FAST = 1: the plain Langevin call (ebm_langevin_chain_f32 / _dev_f32), FAST = 2: the same with diagnostics records (the same chain bit
// for bit: return_diagnostics must not change the samples); FAST = 3: the training forward / backward
// (ebm_mlp_backward_acts_f32).  Everything else at these widths (injected noise, clamps, dim 1 chains) stays on MODE 2.
#include "mlp_wide_body.h"

namespace ebm {
namespace widemlp {

#define EBM_THIN(HTV)                                                                             \
  template <>                                                                                     \
  int launch_thin<HTV>(const WideArgs& a, int fast, hipStream_t st, const char* who) {            \
    return fast == 3 ? launch_variant<HTV, 1, 4, 3>(a, st, who)                                   \
           : fast == 2 ? launch_variant<HTV, 1, 4, 2>(a, st, who) : launch_variant<HTV, 1, 4, 1>(a, st, who); \
  }
EBM_THIN(2) EBM_THIN(4)
#undef EBM_THIN

}  // namespace widemlp
}  // namespace ebm

#ifdef EBM_PHASE_TIMES
// scripts/mlp_phase_times.py on a MODE 4 shape: build THIS file alone with -DEBM_PHASE_TIMES (the log is per translation unit)
extern "C" __attribute__((visibility("default"))) int ebm_debug_phase_log(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ebm::widemlp::ebm_phase_log), (size_t)n * sizeof(unsigned long long));
}
#endif
