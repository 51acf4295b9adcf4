// Register-resident streamed-Ps Gaussian Langevin kernels (dims 132 .. 224): instantiations (gauss_big_body.h has the kernels).
#include "gauss_big_body.h"

namespace ebm {

int launch_gauss_res(int tiles, const gbig::BigArgs& a, hipStream_t st) {
  switch (tiles) {
#ifndef EBM_RES_ONLY8  // (A/B builds of scripts/ab_build.sh: one width, a quarter of the compile time)
    case 5: return gbig::launch_res<5>(a, st);
    case 6: return gbig::launch_res<6>(a, st);
    case 7: return gbig::launch_res<7>(a, st);
#endif
    case 8: return gbig::launch_res<8>(a, st);  // (the caller checked: the image is there -- without it eight tiles run tiled)
    default: return fail(EBM_EDIM, "ebm_langevin_chain_f32: the register-resident Gaussian kernel takes 5 .. 8 tiles, not %d", tiles);
  }
}

}  // namespace ebm
