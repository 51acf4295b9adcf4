// The IMG instantiations of the tiled dense-Gaussian Langevin kernel (gauss_big_body.h: slabs of Ps arrive ready-made by
// LDS-direct loads) and the builder of the image they read.  Reference: torchebm/core/base_model.py:181-210 (GaussianModel:
// the gradient cov_inv (x - mean) is what every step contracts), samplers/langevin_dynamics.py:150-185 (the step loop).
#include "gauss_big_body.h"

namespace ebm {
namespace gbig {

// One thread per lane-operand unit: eight consecutive fp32 of a row of Ps -> three bf16x8 pieces at
// [slice][stage][piece][unit] (unit = [out tile][K-block of the stage][lane (row in tile, K-half)]), zero beyond dim.
// lo: the image of the SHIFTED matrix P'[r][c] = Ps[r - lo][c - lo] (zero outside), one per alignment class of a width that is
// not a multiple of 4 (gauss_mfma_body.h SH); 0: Ps itself.
__global__ __launch_bounds__(256) void gauss_prec_image_kernel(const float* __restrict__ prec, int dim, int ot_n, int kbs, int ns, int n_stage,
                                                               char* __restrict__ out, int lo) {
  const int units = ot_n * 64 * kbs;
  const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (g >= (int64_t)ns * n_stage * units) return;
  const int u = (int)(g % units), s = (int)((g / units) % n_stage), sl = (int)(g / ((int64_t)units * n_stage));
  const int it = u / (64 * kbs), kb2 = (u >> 6) % kbs, ul = u & 63;
  const int row = sl * 32 * ot_n + 32 * it + (ul & 31) - lo;
  const int kcol = 16 * kbs * s + 16 * kb2 + 8 * (ul >> 5) - lo;
  f32x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = (row >= 0 && row < dim && kcol + i >= 0 && kcol + i < dim) ? prec[(int64_t)row * dim + kcol + i] : 0.0f;
  const Tri t = split8(v);
  bf16x8* dst = reinterpret_cast<bf16x8*>(out) + ((int64_t)sl * n_stage + s) * 3 * units + u;
  dst[0] = t.h; dst[units] = t.m; dst[2 * units] = t.l;
}

// The resident kernel's layout (gauss_res_langevin_kernel: units of a (tile, K-block, K-half) group rotated by 2 (2 kb2 + h')):
// one thread per unit of [stage][piece][tile j][kb2][h'][slot]; the unit of row r holds Ps[32 j + r][32 s + 16 kb2 + 4 h' + {0..3, 8..11}].
__global__ __launch_bounds__(256) void gauss_prec_image_res_kernel(const float* __restrict__ prec, int dim, int ot_n, char* __restrict__ out, int lo) {
  const int slabu = ot_n * 128;
  const int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= ot_n * slabu) return;
  const int u = g % slabu, s = g / slabu;
  const int j = u >> 7, kb2 = (u >> 6) & 1, hh = (u >> 5) & 1, slot = u & 31;
  const int r = (slot - 2 * (2 * kb2 + hh)) & 31;
  const int row = 32 * j + r - lo, k0 = 32 * s + 16 * kb2 + 4 * hh - lo;
  f32x8 v;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int col = k0 + (i & 3) + 8 * (i >> 2);
    v[i] = (row >= 0 && row < dim && col >= 0 && col < dim) ? prec[(int64_t)row * dim + col] : 0.0f;
  }
  const Tri t = split8(v);
  bf16x8* dst = reinterpret_cast<bf16x8*>(out) + (int64_t)s * 3 * slabu + u;
  dst[0] = t.h; dst[slabu] = t.m; dst[2 * slabu] = t.l;
}

#define EBM_BIG_IMG(OTV, NSV) \
  template <> int launch_big_img<OTV, NSV>(const BigArgs& a, hipStream_t st) { return launch_big<OTV, NSV, true>(a, st); }
EBM_BIG_IMG(5, 1) EBM_BIG_IMG(6, 1) EBM_BIG_IMG(7, 1) EBM_BIG_IMG(8, 1)
EBM_BIG_IMG(5, 2) EBM_BIG_IMG(6, 2) EBM_BIG_IMG(7, 2) EBM_BIG_IMG(8, 2)
#undef EBM_BIG_IMG

namespace {
struct ImgShape {
  int ot, ns, kbs;
  size_t bytes;      // the tiled kernel's image ...
  size_t res_bytes;  // ... and behind it the resident kernel's (widths it takes: up to eight tiles), or 0
};
// the (OT, NS) the dispatch of gauss_big.hip picks for this width
template <int OT, int NS>
ImgShape shape_of(int dim) {
  constexpr bool kRes = NS == 1;
  return ImgShape{OT, NS, BigCfg<OT, NS>::KBS, big_image_bytes<OT, NS>(dim), kRes ? res_image_bytes<OT>() : 0};
}
ImgShape image_shape(int32_t dim) {
  const int tiles = (dim + 31) / 32;
  if (tiles <= 8) {
    switch (tiles) {
      case 5: return shape_of<5, 1>(dim);
      case 6: return shape_of<6, 1>(dim);
      case 7: return shape_of<7, 1>(dim);
      default: return shape_of<8, 1>(dim);
    }
  }
  switch ((tiles + 1) / 2) {
    case 5: return shape_of<5, 2>(dim);
    case 6: return shape_of<6, 2>(dim);
    case 7: return shape_of<7, 2>(dim);
    default: return shape_of<8, 2>(dim);
  }
}
}  // namespace
}  // namespace gbig

bool gauss_big_supported(int32_t dim);  // gauss_big.hip

// ebm_gauss_prec_image_bytes / ebm_gauss_prec_image_f32 (api.hip)
// Widths off multiples of 4 whose shifted rows reach 161 .. 256 tile coordinates (dim 158 / 159 .. 253 / 254): one image per
// alignment class, each that of a full (32 tiles)^2 matrix -- [class][tiled | resident]; the streamed HMC evaluation on
// shifted rows reads them (gauss_hmc_stream_shift.hip).
static int shift_classes(int32_t dim) { return (dim & 1) ? 4 : 2; }
static int32_t shift_extent(int32_t dim) { return dim + ((dim & 1) ? 3 : 2); }
bool gauss_stream_shift_dim(int32_t dim) { return (dim % 4) != 0 && shift_extent(dim) > 160 && shift_extent(dim) <= 256; }
size_t gauss_prec_image_class_bytes(int32_t dim) {  // one class's share
  const gbig::ImgShape sh = gbig::image_shape(32 * ((shift_extent(dim) + 31) / 32));
  return sh.bytes + sh.res_bytes;
}

size_t gauss_prec_image_bytes(int32_t dim) {
  if (gauss_stream_shift_dim(dim)) return shift_classes(dim) * gauss_prec_image_class_bytes(dim);
  if (!gauss_big_supported(dim)) return 0;
  const gbig::ImgShape sh = gbig::image_shape(dim);
  return sh.bytes + sh.res_bytes;
}
int launch_gauss_prec_image(const float* prec, int32_t dim, void* image, hipStream_t st, const char* who) {
  if (gauss_stream_shift_dim(dim)) {
    if (!prec || !image) return fail(EBM_EINVAL, "%s: NULL pointer", who);
    if (reinterpret_cast<uintptr_t>(image) & 15) return fail(EBM_EINVAL, "%s: the image must be 16-byte aligned", who);
    const int full = 32 * ((shift_extent(dim) + 31) / 32);
    const gbig::ImgShape sh = gbig::image_shape(full);
    const size_t stride = sh.bytes + sh.res_bytes;
    const int n_stage = full / (16 * sh.kbs);
    const int64_t work = (int64_t)sh.ns * n_stage * sh.ot * 64 * sh.kbs;
    const int res_work = sh.ot * sh.ot * 128;
    for (int s = 0; s < shift_classes(dim); ++s) {
      const int lo = (dim * s) & 3;
      char* dst = static_cast<char*>(image) + (size_t)s * stride;
      hipLaunchKernelGGL(gbig::gauss_prec_image_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, prec, dim, sh.ot, sh.kbs, sh.ns,
                         n_stage, dst, lo);
      hipLaunchKernelGGL(gbig::gauss_prec_image_res_kernel, dim3((unsigned)((res_work + 255) / 256)), dim3(256), 0, st, prec, dim, sh.ot,
                         dst + sh.bytes, lo);
    }
    return check_launch(who);
  }
  if (!gauss_big_supported(dim)) return fail(EBM_EDIM, "%s: no precision image at dim %d (132 .. 512 in steps of 4 only)", who, dim);
  if (!prec || !image) return fail(EBM_EINVAL, "%s: NULL pointer", who);
  if (reinterpret_cast<uintptr_t>(image) & 15) return fail(EBM_EINVAL, "%s: the image must be 16-byte aligned", who);
  const gbig::ImgShape sh = gbig::image_shape(dim);
  const int n_stage = ((dim + 31) & ~31) / (16 * sh.kbs);
  const int64_t work = (int64_t)sh.ns * n_stage * sh.ot * 64 * sh.kbs;
  hipLaunchKernelGGL(gbig::gauss_prec_image_kernel, dim3((unsigned)((work + 255) / 256)), dim3(256), 0, st, prec, dim, sh.ot, sh.kbs, sh.ns, n_stage,
                     static_cast<char*>(image), 0);
  if (sh.res_bytes) {
    const int res_work = sh.ot * sh.ot * 128;
    hipLaunchKernelGGL(gbig::gauss_prec_image_res_kernel, dim3((unsigned)((res_work + 255) / 256)), dim3(256), 0, st, prec, dim, sh.ot,
                       static_cast<char*>(image) + sh.bytes, 0);
  }
  return check_launch(who);
}

}  // namespace ebm
