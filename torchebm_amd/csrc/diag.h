// In-kernel sampler diagnostics (samplers/langevin_dynamics.py:170-185, samplers/hmc.py:294-310 of the
// reference: per kept step the population mean / biased variance of every coordinate, the mean energy and,
// for HMC, the acceptance rate).
//
// The chain kernels keep the state in registers for all k steps, so the statistics are taken where the state
// is: at a kept step every workgroup reduces ITS chains to one record of per-column partials and stores it;
// a small finishing kernel (misc.hip: diag_finish_kernel) merges the records.  No atomics and no second pass
// over the state: return_diagnostics=True stays one chain launch.
//
// Geometry.  A workgroup covers E consecutive flat elements of the row-major chain matrix (E = 1024 for the
// flat element-wise kernel, chains_per_block * dim for the lane-group kernels) with either E % dim == 0
// (whole rows) or dim % E == 0 (a row is several blocks).  S = min(dim, E) slots; slot s collects the block's
// elements s, s + dim, s + 2 dim, ... (the block's rows of one column; exactly one element when dim >= E).
//
// Record of block b at kept step j:  rec = partials + (j * n_blocks + b) * (2 S + 2)
//   rec[s]        sum of the slot's elements
//   rec[S + s]    M2 = sum (x - slot mean)^2      (two-pass inside the block: no cancellation)
//   rec[2 S]      sum of the block's per-chain energies      rec[2 S + 1]  number of accepted chains (HMC)
// The finishing kernel merges (count, sum, M2) triples with the pairwise-variance identity in fp64, so the
// result does not depend on how far the population mean is from zero.  Deterministic: same launch, same bits.
#pragma once
#include "ebm_common.h"

namespace ebm {
namespace diag {

struct DiagArgs {
  float* partials;   // null: diagnostics off
  int32_t S;         // slots per block
  int32_t E;         // flat elements per block
  int64_t n_blocks;  // records per kept step
};

constexpr int kBlock = 256;
constexpr int kMaxPart = 16;  // row partitions per slot (small dims: several threads share a slot)

__host__ __device__ inline int scratch_floats(int S) { return 2 * (S > kBlock ? S : kBlock) + 8; }
// LDS floats emit() needs behind the caller's own LDS: the tile, two scratch rows, eight wave partials
__host__ __device__ inline int lds_floats(int E, int S) { return E + scratch_floats(S); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
  return v;
}

// Called by ALL threads of the workgroup at a kept step, after each thread has written its elements of the
// block's state into tile[0 .. E) in flat order (element offset inside the block).  L = number of valid flat
// elements of this block (a multiple of dim when E % dim == 0).  e_part / acc_part: this thread's share of the
// block's energy sum and accept count.  Ends with a barrier: the tile may be rewritten right after.
__device__ __noinline__ void emit(const DiagArgs d, int keep, float* tile, int L, int dim, float e_part, float acc_part) {
  const int tid = threadIdx.x;
  const int S = d.S;
  const int smax = S > kBlock ? S : kBlock;
  float* sum_s = tile + d.E;        // [P][S]
  float* m2_s = sum_s + smax;       // [P][S]
  float* red = m2_s + smax;         // [8]
  e_part = wave_sum(e_part);
  acc_part = wave_sum(acc_part);
  if ((tid & 63) == 0) {
    red[tid >> 6] = e_part;
    red[4 + (tid >> 6)] = acc_part;
  }
  const int SP = S < kBlock ? S : kBlock;
  int P = kBlock / SP;
  if (P > kMaxPart) P = kMaxPart;
  const int p = tid / SP;
  const bool worker = p < P;
  const int s0 = tid - p * SP;
  __syncthreads();  // tile + red written
  if (worker) {
    for (int s = s0; s < S; s += SP) {
      float acc = 0.0f;
      for (int o = s + p * dim; o < L; o += P * dim) acc += tile[o];
      sum_s[p * S + s] = acc;
    }
  }
  __syncthreads();
  if (worker) {
    for (int s = s0; s < S; s += SP) {
      float tot = 0.0f;
      for (int q = 0; q < P; ++q) tot += sum_s[q * S + s];
      const int cnt = s < L ? (L - s + dim - 1) / dim : 0;
      const float mean = cnt > 0 ? tot / (float)cnt : 0.0f;
      float acc = 0.0f;
      for (int o = s + p * dim; o < L; o += P * dim) {
        const float dv = tile[o] - mean;
        acc = __builtin_fmaf(dv, dv, acc);
      }
      m2_s[p * S + s] = acc;
    }
  }
  __syncthreads();
  float* rec = d.partials + ((int64_t)keep * d.n_blocks + blockIdx.x) * (int64_t)(2 * S + 2);
  if (p == 0) {
    for (int s = s0; s < S; s += SP) {
      float tot = 0.0f, m2 = 0.0f;
      for (int q = 0; q < P; ++q) {
        tot += sum_s[q * S + s];
        m2 += m2_s[q * S + s];
      }
      rec[s] = tot;
      rec[S + s] = m2;
    }
  }
  if (tid == 0) {
    rec[2 * S] = (red[0] + red[1]) + (red[2] + red[3]);
    rec[2 * S + 1] = (red[4] + red[5]) + (red[6] + red[7]);
  }
  __syncthreads();  // everyone is done with the tile and the scratch rows
}

// host side: the layout of a launch that covers `block_elems` flat elements per workgroup
inline bool plan(int64_t n_chains, int32_t dim, int64_t block_elems, DiagArgs& d) {
  if (block_elems <= 0) return false;
  if (block_elems % dim != 0 && dim % block_elems != 0) return false;
  d.E = (int32_t)block_elems;
  d.S = dim < block_elems ? dim : (int32_t)block_elems;
  d.n_blocks = ceil_div64(n_chains * (int64_t)dim, block_elems);
  return true;
}

}  // namespace diag
}  // namespace ebm
