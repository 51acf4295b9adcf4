// Micro-benchmark: per-instruction issue cost on gfx950 of the ops the RNG is made of.
// Each kernel runs a long dependent-free stream of one op on 8 independent registers per lane,
// 256 CUs x 8 waves/SIMD, and reports wave-instructions per cycle per SIMD (at the measured time).
//   hipcc --offload-arch=gfx950 -O3 scripts/ubench/valu_rates.hip -o gpurun_out/valu_rates && gpurun_out/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int kIters = 4096;
constexpr int kUnroll = 8;

#define BODY_LOOP(STMT)                                   \
  for (int it = 0; it < kIters; ++it) {                   \
    _Pragma("unroll") for (int j = 0; j < kUnroll; ++j) { STMT; } \
  }

__global__ void k_fma(float* out) {
  float a[kUnroll];
  for (int j = 0; j < kUnroll; ++j) a[j] = threadIdx.x * 1e-3f + j;
  BODY_LOOP(a[j] = __builtin_fmaf(a[j], 1.0001f, 0.5f))
  float s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_xor(float* out) {
  uint32_t a[kUnroll];
  for (int j = 0; j < kUnroll; ++j) a[j] = threadIdx.x * 2654435761u + j;
  BODY_LOOP(a[j] = (a[j] ^ 0x9E3779B9u) + 1u)   // xor + add: 2 full-rate ops
  uint32_t s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}
__global__ void k_mad64(float* out) {
  uint32_t a[kUnroll];
  for (int j = 0; j < kUnroll; ++j) a[j] = threadIdx.x * 2654435761u + j;
  BODY_LOOP({ uint64_t p = (uint64_t)0xD2511F53u * a[j]; a[j] = (uint32_t)(p >> 32) ^ (uint32_t)p; })  // mad_u64 + xor
  uint32_t s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}
__global__ void k_mullo(float* out) {
  uint32_t a[kUnroll];
  for (int j = 0; j < kUnroll; ++j) a[j] = threadIdx.x * 2654435761u + j;
  BODY_LOOP(a[j] = a[j] * 0xD2511F53u + 1u)
  uint32_t s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}
__global__ void k_mulhi(float* out) {
  uint32_t a[kUnroll];
  for (int j = 0; j < kUnroll; ++j) a[j] = threadIdx.x * 2654435761u + j;
  BODY_LOOP(a[j] = __umulhi(a[j], 0xD2511F53u) + 0x12345u)
  uint32_t s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}
__global__ void k_mul24(float* out) {
  uint32_t a[kUnroll];
  for (int j = 0; j < kUnroll; ++j) a[j] = threadIdx.x * 2654435761u + j;
  BODY_LOOP(a[j] = __umul24(a[j], 0x511F53u) + 3u)
  uint32_t s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}
#define TRANS_KERNEL(NAME, EXPR)                                            \
  __global__ void NAME(float* out) {                                        \
    float a[kUnroll];                                                       \
    for (int j = 0; j < kUnroll; ++j) a[j] = 0.3f + threadIdx.x * 1e-4f + j * 0.01f; \
    BODY_LOOP(a[j] = EXPR)                                                  \
    float s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j];               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;                         \
  }
TRANS_KERNEL(k_log, __builtin_amdgcn_logf(a[j]) + 2.0f)
TRANS_KERNEL(k_sin, __builtin_amdgcn_sinf(a[j]) + 0.7f)
TRANS_KERNEL(k_sqrt, __builtin_amdgcn_sqrtf(a[j]) + 0.1f)
TRANS_KERNEL(k_exp, __builtin_amdgcn_exp2f(a[j]) * 0.3f)
TRANS_KERNEL(k_rcp, __builtin_amdgcn_rcpf(a[j]) + 0.2f)
// transcendental + independent fma stream: do they overlap?
__global__ void k_log_fma(float* out) {
  float a[kUnroll], b[kUnroll];
  for (int j = 0; j < kUnroll; ++j) { a[j] = 0.3f + threadIdx.x * 1e-4f + j * 0.01f; b[j] = a[j]; }
  BODY_LOOP({ a[j] = __builtin_amdgcn_logf(a[j]) + 2.0f; b[j] = __builtin_fmaf(b[j], 1.0001f, 0.5f); b[j] = __builtin_fmaf(b[j], 0.9999f, 0.25f); })
  float s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j] + b[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ void k_mad64_fma(float* out) {
  uint32_t a[kUnroll]; float b[kUnroll];
  for (int j = 0; j < kUnroll; ++j) { a[j] = threadIdx.x * 2654435761u + j; b[j] = j + 0.5f; }
  BODY_LOOP({ uint64_t p = (uint64_t)0xD2511F53u * a[j]; a[j] = (uint32_t)(p >> 32) ^ (uint32_t)p; b[j] = __builtin_fmaf(b[j], 1.0001f, 0.5f); b[j] = __builtin_fmaf(b[j], 0.9999f, 0.25f); })
  uint32_t s = 0; float t = 0; for (int j = 0; j < kUnroll; ++j) { s += a[j]; t += b[j]; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s) + t;
}

// gfx950's three-input logic op (truth table 0x96 = a ^ b ^ c) + add, and a full Philox-round-shaped
// mix: 2 x (mad_u64 + bitop3)
__global__ void k_bitop3(float* out) {
  uint32_t a[kUnroll];
  for (int j = 0; j < kUnroll; ++j) a[j] = threadIdx.x * 2654435761u + j;
  BODY_LOOP(a[j] = __builtin_amdgcn_bitop3_b32(a[j], 0x9E3779B9u, (uint32_t)threadIdx.x, 0x96) + 1u)
  uint32_t s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}
__global__ void k_philox_round(float* out) {
  uint32_t a[kUnroll], b[kUnroll];
  for (int j = 0; j < kUnroll; ++j) { a[j] = threadIdx.x * 2654435761u + j; b[j] = a[j] ^ 0x55555555u; }
  BODY_LOOP({ uint64_t p = (uint64_t)0xD2511F53u * a[j]; uint64_t q = (uint64_t)0xCD9E8D57u * b[j];
              a[j] = __builtin_amdgcn_bitop3_b32((uint32_t)(q >> 32), (uint32_t)p, 0x9E3779B9u, 0x96);
              b[j] = __builtin_amdgcn_bitop3_b32((uint32_t)(p >> 32), (uint32_t)q, 0xBB67AE85u, 0x96); })
  uint32_t s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j] + b[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = __uint_as_float(s);
}

// packed fp32 FMA (2 FMAs per lane per instruction)
__global__ void k_pk_fma(float* out) {
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f a[kUnroll];
  for (int j = 0; j < kUnroll; ++j) a[j] = v2f{threadIdx.x * 1e-3f + j, 0.5f + j};
  BODY_LOOP(a[j] = __builtin_elementwise_fma(a[j], v2f{1.0001f, 0.9999f}, v2f{0.5f, 0.25f}))
  float s = 0; for (int j = 0; j < kUnroll; ++j) s += a[j].x + a[j].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
void run(const char* name, K kern, float* out, double ops_per_iter) {
  const int blocks = 256 * 8, threads = 256;
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), 0, 0, out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  ms /= 5;
  const double wave_instr = (double)blocks * (threads / 64) * kIters * kUnroll * ops_per_iter;
  const double per_simd_per_s = wave_instr / (ms * 1e-3) / (256.0 * 4);
  // cycles per wave-instruction per SIMD at 2.4 GHz nominal
  printf("%-14s %8.3f ms   %.3f ns per wave-instr per SIMD  = %.2f cycles @2.4GHz (ops counted: %.0f/iter)\n", name, ms,
         1e9 / per_simd_per_s, 2.4 * 1e9 / per_simd_per_s, ops_per_iter);
}

int main() {
  float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
  run("fma", k_fma, out, 1);
  run("xor+add", k_xor, out, 2);
  run("mad64+xor", k_mad64, out, 2);
  run("mul_lo+add", k_mullo, out, 2);
  run("mul_hi+add", k_mulhi, out, 2);
  run("mul_u24+add", k_mul24, out, 2);
  run("log+add", k_log, out, 2);
  run("sin+add", k_sin, out, 2);
  run("sqrt+add", k_sqrt, out, 2);
  run("exp+mul", k_exp, out, 2);
  run("rcp+add", k_rcp, out, 2);
  run("log+add+2fma", k_log_fma, out, 4);
  run("mad64+xor+2fma", k_mad64_fma, out, 4);
  run("pk_fma", k_pk_fma, out, 1);
  run("bitop3+add", k_bitop3, out, 2);
  run("philox round (2 mad64 + 2 bitop3)", k_philox_round, out, 4);
  return 0;
}
