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
  // ... with the sampler diagnostics of the last step taken inside the launch (thin = k: one kept step):
  // per-workgroup records, merged by ebm_diag_finish_f32 (samplers/langevin_dynamics.py:170-185)
  int64_t n_blocks = 0;
  int32_t slots = 0, block_elems = 0;
  EBM_OK(ebm_diag_layout(&e, EBM_DIAG_LANGEVIN, n, dim, 0, 0, &n_blocks, &slots, &block_elems));
  float *records, *d_mean, *d_var, *d_energy;
  double* work;
  HIP_OK(hipMalloc(&records, (size_t)n_blocks * (2 * slots + 8) * sizeof(float)));
  HIP_OK(hipMalloc(&d_mean, dim * sizeof(float)));
  HIP_OK(hipMalloc(&d_var, dim * sizeof(float)));
  HIP_OK(hipMalloc(&d_energy, sizeof(float)));
  HIP_OK(hipMalloc(&work, (3 * dim + 3) * sizeof(double)));
  HIP_OK(hipMemsetAsync(work, 0, (3 * dim + 3) * sizeof(double), st));
  EBM_OK(ebm_langevin_chain_f32(&e, xa, n, dim, k, eta, sqrt_eta, coef, nullptr, 0, 0.0f, 0.0f, k, nullptr, records, nullptr,
                                seed, step0, st));
  EBM_OK(ebm_diag_finish_f32(records, 1, n_blocks, slots, block_elems, n, dim, d_mean, d_var, d_energy, nullptr, work, st));
  // (b) the same field written out, then the injected-noise form of the same entry point
  for (int32_t s = 0; s < k; ++s)
    EBM_OK(ebm_noise_fill_f32(noise + (size_t)s * n * dim, n * dim, EBM_NOISE_NORMAL, seed, step0 + (uint64_t)s, st));
  EBM_OK(ebm_langevin_chain_f32(&e, xb, n, dim, k, eta, sqrt_eta, coef, nullptr, 0, 0.0f, 0.0f, 1, nullptr, nullptr, noise, 0, 0,
                                st));
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
  // the in-kernel diagnostics against the same statistics computed on the host from the final state
  std::vector<float> h_mean(dim), h_var(dim);
  float h_energy = 0.0f;
  HIP_OK(hipMemcpy(h_mean.data(), d_mean, dim * sizeof(float), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(h_var.data(), d_var, dim * sizeof(float), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(&h_energy, d_energy, sizeof(float), hipMemcpyDeviceToHost));
  double worst = 0.0, e_ref = 0.0;
  for (int32_t c = 0; c < dim; ++c) {
    double s1 = 0.0, s2 = 0.0;
    for (int64_t r = 0; r < n; ++r) s1 += a[(size_t)r * dim + c];
    const double m = s1 / (double)n;
    for (int64_t r = 0; r < n; ++r) { const double d = a[(size_t)r * dim + c] - m; s2 += d * d; }
    worst = std::fmax(worst, std::fabs(m - h_mean[c]));
    worst = std::fmax(worst, std::fabs(s2 / (double)n - h_var[c]) / (s2 / (double)n));
  }
  for (size_t i = 0; i < a.size(); ++i) { const double u = (double)a[i] * a[i] - 1.0; e_ref += 2.0 * u * u; }
  e_ref /= (double)n;
  const bool diag_ok = worst < 1e-4 && std::fabs(e_ref - h_energy) < 1e-4 * std::fabs(e_ref);
  std::printf("in-kernel diagnostics: worst mean/var deviation %.2e, energy %.5f (host %.5f) -> %s\n", worst, h_energy, e_ref,
              diag_ok ? "ok" : "MISMATCH");
  hipFree(records); hipFree(d_mean); hipFree(d_var); hipFree(d_energy); hipFree(work);

  // argument errors come back as codes + a message, never as exceptions
  const int rc = ebm_langevin_chain_f32(&e, nullptr, n, dim, k, eta, sqrt_eta, coef, nullptr, 0, 0.0f, 0.0f, 1, nullptr,
                                        nullptr, nullptr, seed, step0, st);
  std::printf("NULL state -> rc %d (%s)\n", rc, ebm_last_error_string());

  hipFree(xa); hipFree(xb); hipFree(noise);
  hipStreamDestroy(st);
  const bool ok = diag_ok && differing == 0 && nonfinite == 0 && rc == EBM_EINVAL && mean_abs > 0.5 && mean_abs < 1.5;
  std::printf(ok ? "c_abi_demo: OK\n" : "c_abi_demo: FAILED\n");
  return ok ? 0 : 1;
}
