// FAST = 3 instantiations of the wide-MLP kernel (MODE 2 shapes up to dim 64): the evaluation as the BACKWARD pass of a training
// step -- ebm_mlp_backward_acts_f32: seed-scaled backward through the network with the four activations the parameter
// gradients are made of stored hidden-major (mlp_wide_eval_b16.inc, `eval_store_acts`).  Reference: what autograd does for
// loss.backward() through torchebm/losses/contrastive_divergence.py:128-155 on the network of
// examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-31.
#include "mlp_wide_body.h"

namespace ebm {
namespace widemlp {

#define EBM_TRAIN(HTV, DTV)                                                                       \
  template <>                                                                                     \
  int launch_train<HTV, DTV>(const WideArgs& a, hipStream_t st, const char* who) {                \
    return launch_variant<HTV, DTV, 2, 3>(a, st, who);                                            \
  }
EBM_TRAIN(2, 1) EBM_TRAIN(2, 2) EBM_TRAIN(4, 1) EBM_TRAIN(4, 2)
#undef EBM_TRAIN

}  // namespace widemlp

// hidden 64 / 128, dim <= 64
int launch_mlp_backward_acts(int32_t hidden, const float* params, const float* x, int64_t n_chains, int32_t dim, const float* seed,
                             float* energy_out, float* grad_out, float* acts, hipStream_t st, const char* who) {
  using namespace widemlp;
  WideArgs a{};
  a.x = const_cast<float*>(x); a.n_chains = n_chains; a.dim = dim; a.k_steps = 0;
  a.thin = 1; a.params = params; a.energy_out = energy_out; a.grad_out = grad_out; a.seed = seed; a.acts = acts; a.act_stride = (n_chains + 127) / 128 * 128;  // whole workgroups of 4 x 32 chains: no lane, no wave needs masking
  a.diag_blocks = ceil_div64(n_chains, 32);
  const int dt = (dim + 31) / 32;
#ifndef EBM_MLP_NO_THIN
  if (dim <= 2 && !ab_switch("EBM_MLP_NO_THIN")) return hidden == 64 ? launch_thin<2>(a, 3, st, who) : launch_thin<4>(a, 3, st, who);
#endif
  if (hidden == 64) return dt == 1 ? launch_train<2, 1>(a, st, who) : launch_train<2, 2>(a, st, who);
  return dt == 1 ? launch_train<4, 1>(a, st, who) : launch_train<4, 2>(a, st, who);
}

}  // namespace ebm
