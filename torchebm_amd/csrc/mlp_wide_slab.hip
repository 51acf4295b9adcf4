// MODE 3 of the wide-MLP chain kernel (mlp_wide_body.h): hidden width 128 at input widths 65 .. 128 -- the reference's
// benchmark network at its widest (benchmarks/registry.py:372-387, dim 128) -- on the bf16 matrix pipe with three-way split
// operands.  The two weight images together are 192 KB, the CU has 160: the W2 image lives in LDS, the W1 image is built ONCE
// per parameter set in global memory (ebm_mlp_w1_image_f32 below) and walked slab by slab through two 24 KB LDS buffers
// filled by LDS-direct loads one slab ahead of the MFMAs (mlp_b16.h, "MODE 3").  96 KB, read by every workgroup: L2-resident.
#include "mlp_wide_body.h"

namespace ebm {
namespace widemlp {

// the image: slab s (hidden rows 32 s .. 32 s + 31) = three splits of [32][128] in the layout stage_image<32, 128> gives LDS
__global__ __launch_bounds__(256) void mlp_w1_image_kernel(const float* __restrict__ w1, int hidden, int dim, char* __restrict__ out) {
  using namespace mlpb16;
  constexpr int QPR = kSlabCols / 4;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= hidden * QPR) return;
  const int row = i / QPR, cq = i - row * QPR;
  bf16x4 hi, mid, lo;
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int c = 4 * cq + e;
    const float v = c < dim ? w1[(size_t)row * dim + c] : 0.0f;
    __bf16 a, b, c3;
    gauss3::split3(v, a, b, c3);
    hi[e] = a; mid[e] = b; lo[e] = c3;
  }
  constexpr uint32_t SPLIT = (uint32_t)kSlabRows * 2u * kSlabCols;
  char* slab = out + (size_t)(row / kSlabRows) * kSlabBytes + Img<kSlabCols>::atom((uint32_t)(row % kSlabRows), (uint32_t)cq);
  *reinterpret_cast<bf16x4*>(slab) = hi;
  *reinterpret_cast<bf16x4*>(slab + SPLIT) = mid;
  *reinterpret_cast<bf16x4*>(slab + 2u * SPLIT) = lo;
}

template <int DT>
static int launch_slab_impl(const WideArgs& a, int fast, hipStream_t st, const char* who) {
  switch (fast) {
    case 1: return launch_variant<4, DT, 3, 1>(a, st, who);
    case 2: return launch_variant<4, DT, 3, 2>(a, st, who);
    default: return launch_variant<4, DT, 3, 0>(a, st, who);
  }
}
template <> int launch_slab<3>(const WideArgs& a, int fast, hipStream_t st, const char* who) { return launch_slab_impl<3>(a, fast, st, who); }
template <> int launch_slab<4>(const WideArgs& a, int fast, hipStream_t st, const char* who) { return launch_slab_impl<4>(a, fast, st, who); }

}  // namespace widemlp

// ebm_mlp_w1_image_bytes / ebm_mlp_w1_image_f32 (api.hip)
size_t mlp_w1_image_bytes(int32_t hidden, int32_t dim) {
  return (hidden == 128 && dim > 64 && dim <= 128) ? (size_t)(hidden / mlpb16::kSlabRows) * mlpb16::kSlabBytes : 0;
}
int launch_mlp_w1_image(const float* params, int32_t hidden, int32_t dim, void* image, hipStream_t st, const char* who) {
  if (mlp_w1_image_bytes(hidden, dim) == 0) return fail(EBM_EDIM, "%s: no W1 image for hidden width %d at dim %d (128 at 65 .. 128 only)", who, hidden, dim);
  if (!params || !image) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  if (reinterpret_cast<uintptr_t>(image) & 15) return fail(EBM_EINVAL, "%s: the image must be 16-byte aligned", who);
  const int work = hidden * (mlpb16::kSlabCols / 4);
  hipLaunchKernelGGL(widemlp::mlp_w1_image_kernel, dim3((work + 255) / 256), dim3(256), 0, st, params, hidden, dim, static_cast<char*>(image));
  return check_launch(who);
}

}  // namespace ebm

#ifdef EBM_PHASE_TIMES
// scripts/mlp_phase_times.py on a MODE 3 shape: build THIS file alone with -DEBM_PHASE_TIMES (the log is per translation unit)
extern "C" __attribute__((visibility("default"))) int ebm_debug_phase_log(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ebm::widemlp::ebm_phase_log), (size_t)n * sizeof(unsigned long long));
}
#endif
