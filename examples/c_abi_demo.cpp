// The C ABI without Python or PyTorch: a plain HIP host program that links libebm_hip.so, runs the fused
// Langevin chain on a DoubleWell energy with the library's own Philox noise, then reproduces the same
// chain from the materialised noise field (ebm_noise_fill_f32 + the injected-noise form) and checks that
// the two agree bit for bit -- the property the Python tests pin, shown at the boundary itself.
//
//   hipcc --offload-arch=gfx950 -I include examples/c_abi_demo.cpp -L torchebm_amd -lebm_hip \
//         -Wl,-rpath,$PWD/torchebm_amd -o build/c_abi_demo && build/c_abi_demo
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "ebm_hip.h"

#define HIP_OK(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      std::fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));          \
      return 2;                                                                       \
    }                                                                                 \
  } while (0)
#define EBM_OK(call)                                                                  \
  do {                                                                                \
    int r_ = (call);                                                                  \
    if (r_ != 0) {                                                                    \
      std::fprintf(stderr, "%s failed (%d): %s\n", #call, r_, ebm_last_error_string()); \
      return 3;                                                                       \
    }                                                                                 \
  } while (0)

int main() {
  const int64_t n = 4099;  // not a multiple of anything convenient
  const int32_t dim = 8, k = 12;
  const float eta = 0.01f, sigma = 1.0f;
  const float sqrt_eta = (float)std::sqrt((double)eta), coef = (float)std::sqrt(2.0 * sigma * sigma);
  const uint64_t seed = 1234, step0 = 77;
  std::printf("libebm_hip ABI version %d\n", ebm_version());

  std::vector<float> x0((size_t)n * dim);
  uint32_t lcg = 12345u;
  for (float& v : x0) {  // any deterministic start in [-2, 2)
    lcg = lcg * 1664525u + 1013904223u;
    v = (float)(lcg >> 8) * (4.0f / 16777216.0f) - 2.0f;
  }
  const size_t bytes = x0.size() * sizeof(float);
  float *xa = nullptr, *xb = nullptr, *noise = nullptr;
  HIP_OK(hipMalloc(&xa, bytes));
  HIP_OK(hipMalloc(&xb, bytes));
  HIP_OK(hipMalloc(&noise, bytes * k));
  HIP_OK(hipMemcpy(xa, x0.data(), bytes, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(xb, x0.data(), bytes, hipMemcpyHostToDevice));
  hipStream_t st;
  HIP_OK(hipStreamCreate(&st));

  ebm_energy_t e;
  std::memset(&e, 0, sizeof e);
  e.kind = EBM_ENERGY_DOUBLE_WELL;
  e.s[0] = 2.0f;  // barrier height h
  e.s[1] = 1.0f;  // b * b

  // (a) k fused steps, in-kernel Philox draws at steps step0 .. step0 + k - 1
  EBM_OK(ebm_langevin_chain_f32(&e, xa, n, dim, k, eta, sqrt_eta, coef, nullptr, 0, 0.0f, 0.0f, 1, nullptr, nullptr,
                                seed, step0, st));
  // (b) the same field written out, then the injected-noise form of the same entry point
  for (int32_t s = 0; s < k; ++s)
    EBM_OK(ebm_noise_fill_f32(noise + (size_t)s * n * dim, n * dim, EBM_NOISE_NORMAL, seed, step0 + (uint64_t)s, st));
  EBM_OK(ebm_langevin_chain_f32(&e, xb, n, dim, k, eta, sqrt_eta, coef, nullptr, 0, 0.0f, 0.0f, 1, nullptr, noise, 0, 0, st));
  HIP_OK(hipStreamSynchronize(st));

  std::vector<float> a(x0.size()), b(x0.size());
  HIP_OK(hipMemcpy(a.data(), xa, bytes, hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(b.data(), xb, bytes, hipMemcpyDeviceToHost));
  double mean_abs = 0.0;
  size_t differing = 0, nonfinite = 0;
  for (size_t i = 0; i < a.size(); ++i) {
    if (std::memcmp(&a[i], &b[i], sizeof(float)) != 0) ++differing;
    if (!std::isfinite(a[i])) ++nonfinite;
    mean_abs += std::fabs((double)a[i]);
  }
  mean_abs /= (double)a.size();
  std::printf("chains %lld x dim %d, %d steps: mean |x| = %.4f, non-finite = %zu, native vs injected differing = %zu\n",
              (long long)n, dim, k, mean_abs, nonfinite, differing);

  // argument errors come back as codes + a message, never as exceptions
  const int rc = ebm_langevin_chain_f32(&e, nullptr, n, dim, k, eta, sqrt_eta, coef, nullptr, 0, 0.0f, 0.0f, 1, nullptr,
                                        nullptr, seed, step0, st);
  std::printf("NULL state -> rc %d (%s)\n", rc, ebm_last_error_string());

  hipFree(xa); hipFree(xb); hipFree(noise);
  hipStreamDestroy(st);
  const bool ok = differing == 0 && nonfinite == 0 && rc == EBM_EINVAL && mean_abs > 0.5 && mean_abs < 1.5;
  std::printf(ok ? "c_abi_demo: OK\n" : "c_abi_demo: FAILED\n");
  return ok ? 0 : 1;
}
