// Small device helpers shared by the one-lane-per-chain HMC kernels of dim 32 (hmc_ring.hip, hmc_gmm32.hip).
#pragma once
#include "hmc_kernel.h"

namespace ebm {
namespace hmc {

typedef float v2f __attribute__((ext_vector_type(2)));

namespace lane {

__device__ __forceinline__ float to_sgpr(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
// NaN-propagating maximum of three (v_maximum3_f32)
__device__ __forceinline__ float max3np(float a, float b, float c) {
  return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float v) { return v2f{v, v}; }

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
constexpr int NV = 8, D = 32, NP = 16;   // float4 vectors, columns, packed pairs of a row

}  // namespace lane
}  // namespace hmc
}  // namespace ebm
