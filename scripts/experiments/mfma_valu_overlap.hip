// Does a gfx950 SIMD run vector instructions under MFMAs?  Four loops, one workgroup of 256 (or 512) threads per CU, timed with events:
//   A: MFMAs only (v_mfma_f32_32x32x16_bf16, four independent accumulators)      B: v_fma_f32 only (eight independent chains)
//   C: one wave does both, interleaved 1 MFMA : R FMAs                             D: two waves per SIMD, one does A's loop, one B's
// Build + run: hipcc --offload-arch=gfx950 -O3 mfma_valu_overlap.hip -o /tmp/ovl && /tmp/ovl
#include <hip/hip_runtime.h>
#include <cstdio>
#include <type_traits>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int R>
__global__ __launch_bounds__(512) void k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = (f32x16)(0.0f);
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(1.0f + threadIdx.x * 1e-3f); b[i] = (__bf16)(0.5f); }
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 1e-3f + i;
  const float c = 1.0001f, d = 1e-3f;
  const auto body = [&](auto mf, auto vf) __attribute__((always_inline)) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if constexpr (decltype(mf)::value) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u & 3], 0, 0, 0);
        if constexpr (decltype(vf)::value) {
#pragma unroll
          for (int q = 0; q < R; ++q) v[q & 7] = __builtin_fmaf(v[q & 7], c, d);
        }
      }
    }
  };
  using T = std::true_type; using F = std::false_type;
  if constexpr (MODE == 0) body(T{}, F{});
  else if constexpr (MODE == 1) body(F{}, T{});
  else if constexpr (MODE == 2) body(T{}, T{});
  else { if (wave < 4) body(T{}, F{}); else body(F{}, T{}); }  // D: the roles are decided once, outside the loops
  float s = 0.0f;
  for (int i = 0; i < 8; ++i) s += v[i];
  for (int i = 0; i < 4; ++i) s += acc[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int R>
float run(int threads, int iters) {
  float* out; hipMalloc(&out, 256 * 512 * sizeof(float));
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, R>), dim3(256), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); hipFree(out);
  return ms / 5;
}

template <int R>
void row() {
  const int iters = 20000;
  const float a = run<0, R>(256, iters), b = run<1, R>(256, iters), c = run<2, R>(256, iters), d = run<3, R>(512, iters);
  const float a2 = run<0, R>(512, iters), b2 = run<1, R>(512, iters);
  printf("R = %2d FMAs per MFMA:  A mfma only %.3f ms   B valu only %.3f ms   C one wave both %.3f ms (A + B = %.3f, max = %.3f)   "
         "D two waves, one each %.3f ms   [A at 2 waves/SIMD %.3f, B at 2 waves/SIMD %.3f]\n", R, a, b, c, a + b, a > b ? a : b, d, a2, b2);
}
int main() { row<2>(); row<4>(); row<8>(); row<16>(); return 0; }
