// Device-side helpers shared by the gfx950 kernels: Philox4x32-10, Box-Muller on the
// hardware transcendental unit, NaN-propagating clamps, error plumbing.
//
// This translation unit is compiled with -ffp-contract=off: every `a*b+c` written with
// plain operators is a rounded multiply followed by a rounded add, which is what the
// reference's eager torch ops do.  FMAs appear only where
// written explicitly (__builtin_fmaf), i.e. inside the RNG.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/ebm_hip.h"

namespace ebm {

// A/B switches (EBM_GAUSS_ROWS=1 and friends) exist only in builds made with -DEBM_AB_SWITCHES (make CXXFLAGS_EXTRA=-DEBM_AB_SWITCHES:
// the comparison runs of scripts/).  The shipped library reads no environment variable and keeps no process-global state.
#ifdef EBM_AB_SWITCHES
}  // namespace ebm
#include <cstdlib>
namespace ebm {
inline bool ab_switch(const char* name, char on = '1') {
  const char* v = getenv(name);
  return v && v[0] == on;
}
inline int ab_int(const char* name) {
  const char* v = getenv(name);
  return v ? atoi(v) : 0;
}
#else
constexpr bool ab_switch(const char*, char = '1') { return false; }
constexpr int ab_int(const char*) { return 0; }
#endif


// ---------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------
void set_error(const char* fmt, ...);  // api.hip
int  fail(int code, const char* fmt, ...);
int  check_launch(const char* what);

// first() is true once per device: function attributes (the > 64 KiB dynamic-LDS opt-in) are per device, and one process
// may drive several (a `static bool` would leave every device after the first without the opt-in).  Lock-free.
struct DeviceOnce {
  std::atomic<uint64_t> seen{0};
  bool first() {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const uint64_t bit = 1ull << (dev & 63);
    return (seen.fetch_or(bit, std::memory_order_relaxed) & bit) == 0;
  }
};

// ---------------------------------------------------------------------------------
// Philox4x32-10 (Salmon et al., SC'11).  Known-answer vectors are checked in
// tests/test_rng.py through ebm_noise_fill_f32(kind=RAW_U32).
// One round = two 32x32->64 multiplies (v_mad_u64_u32) + two 3-input xors (v_xor3_b32);
// the key schedule is wave-uniform and stays on the scalar unit.
// ---------------------------------------------------------------------------------
struct U4 {
  uint32_t x, y, z, w;
};

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
#else
  return a ^ b ^ c;
#endif
}

__device__ __forceinline__ U4 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                            uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    // three-input XOR in ONE instruction: gfx950's v_bitop3_b32 with truth table 0x96 (a ^ b ^ c); the
    // compiler emits two v_xor_b32 for the plain expression -- 40 -> 20 logic ops per Philox call
    const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
    const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return U4{c0, c1, c2, c3};
}

struct RngKey {
  uint32_t k0, k1;  // seed
};

__device__ __forceinline__ U4 philox_at(RngKey key, uint64_t group, uint64_t step) {
  return philox4x32_10((uint32_t)group, (uint32_t)(group >> 32), (uint32_t)step,
                       (uint32_t)(step >> 32), key.k0, key.k1);
}

// (0, 1]: never 0, so the log below is finite; same mapping as cuRAND's uniform.
__device__ __forceinline__ float u01_open_low(uint32_t r) {
  return __builtin_fmaf((float)r, 0x1p-32f, 0x1p-33f);
}
// [0, 1): 24 random mantissa bits (torch.rand's convention on the GPU).
__device__ __forceinline__ float u01_half_open(uint32_t r) { return (float)(r >> 8) * 0x1p-24f; }

// Box-Muller on the transcendental unit: v_log_f32 is log2, v_sin/v_cos take
// revolutions, so 2*pi*u2 needs no range reduction at all.
__device__ __forceinline__ void box_muller(uint32_t a, uint32_t b, float& n0, float& n1) {
  const float u1 = u01_open_low(a);
  const float rev = (float)b * 0x1p-32f;
  // -2 ln(u1) = (-2 ln 2) * log2(u1)
  const float r = __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u1));
  n0 = r * __builtin_amdgcn_sinf(rev);
  n1 = r * __builtin_amdgcn_cosf(rev);
}

struct F4 {
  float v[4];
};

__device__ __forceinline__ F4 normal4_at(RngKey key, uint64_t group, uint64_t step) {
  const U4 o = philox_at(key, group, step);
  F4 n;
  box_muller(o.x, o.y, n.v[0], n.v[1]);
  box_muller(o.z, o.w, n.v[2], n.v[3]);
  return n;
}

// The contracted form (ABI 8, EBM_CHAIN_CONTRACTED): the Box-Muller radius carries the update's whole noise coefficient
// A = noise_coef * sqrt_eta, so the update adds the draw as it is -- one multiply per PAIR of normals where the reference order
// (eps * sqrt_eta, then * noise_coef) spends two per normal.  Same Philox counters; rounding differs in the last bit.
__device__ __forceinline__ F4 scaled_normal4_at(RngKey key, uint64_t group, uint64_t step, float A) {
  const U4 o = philox_at(key, group, step);
  F4 n;
  const float r0 = A * __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u01_open_low(o.x)));
  const float r1 = A * __builtin_amdgcn_sqrtf(-1.38629436111989061883f * __builtin_amdgcn_logf(u01_open_low(o.z)));
  const float v0 = (float)o.y * 0x1p-32f, v1 = (float)o.w * 0x1p-32f;
  n.v[0] = r0 * __builtin_amdgcn_sinf(v0);
  n.v[1] = r0 * __builtin_amdgcn_cosf(v0);
  n.v[2] = r1 * __builtin_amdgcn_sinf(v1);
  n.v[3] = r1 * __builtin_amdgcn_cosf(v1);
  return n;
}

__device__ __forceinline__ uint32_t pick(U4 o, int r) {
  return r == 0 ? o.x : (r == 1 ? o.y : (r == 2 ? o.z : o.w));
}

// ---------------------------------------------------------------------------------
// torch.clamp semantics: NaN passes through (fminf/fmaxf would swallow it).  IEEE-754-2019
// maximum / minimum propagate NaN and are single instructions on gfx950 (v_maximum3_f32 /
// v_minimum3_f32): two ops where compare + select takes four.
// ---------------------------------------------------------------------------------
__device__ __forceinline__ float clamp_nanprop(float v, float lo, float hi) {
  return __builtin_elementwise_minimum(__builtin_elementwise_maximum(v, lo), hi);
}

// torch.nan_to_num_(nan=0.0): NaN -> 0, +inf -> FLT_MAX, -inf -> -FLT_MAX.
__device__ __forceinline__ float nan_to_num0(float v) {
  float r = (v > 3.402823466e+38f) ? 3.402823466e+38f : v;   // +inf
  r = (v < -3.402823466e+38f) ? -3.402823466e+38f : r;       // -inf
  return (v != v) ? 0.0f : r;                                  // NaN
}

__host__ __device__ __forceinline__ int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

__host__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace ebm
