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
// Record of block b at kept step j:  rec = partials + (j * n_blocks + b) * (2 S + 8)
//   rec[s]             sum of the slot's elements
//   rec[S + s]         M2 = sum (x - slot mean)^2      (two-pass inside the block: no cancellation)
//   rec[2 S + 0 .. 3]  four shares of the block's energy sum (one per wave on the fast path; share 0 otherwise)
//   rec[2 S + 4 .. 7]  four shares of the block's accept count (HMC)
// The finishing kernel merges (count, sum, M2) triples with the pairwise-variance identity in fp64, so the
// result does not depend on how far the population mean is from zero.  Deterministic: same launch, same bits.
#pragma once
#include <type_traits>

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

__host__ __device__ inline int record_floats(int S) { return 2 * S + 8; }
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
// `scratch`: scratch_floats(S) floats (directly behind the tile unless the caller keeps the tile elsewhere).
__device__ __noinline__ void emit(const DiagArgs d, int keep, const float* tile, float* scratch, int L, int dim, float e_part,
                                  float acc_part) {
  const int tid = threadIdx.x;
  const int S = d.S;
  const int smax = S > kBlock ? S : kBlock;
  float* sum_s = scratch;           // [P][S]
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
  float* rec = d.partials + ((int64_t)keep * d.n_blocks + blockIdx.x) * (int64_t)record_floats(S);
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
  if (tid < 8) rec[2 * S + tid] = red[tid];  // the four waves' energy shares, then their accept shares
  __syncthreads();  // everyone is done with the tile and the scratch rows
}

// Fast path of the flat element-wise kernel for dim = 4 .. 256, a power of two (1024 % dim == 0).  Every lane drops
// its float4 into an LDS tile, ONE barrier (two alternating tiles), then every wave reduces its own QUARTER OF THE
// COLUMNS over all rows of the tile: wave w owns columns [w dim/4, (w+1) dim/4), 64 / (dim/4) lanes share a
// column and each reads 4 rows; sum, xor-shuffle across the sharing lanes, mean, squared deviations, shuffle
// again; the first dim/4 lanes store the wave's part of the record.  All four waves do the same ~70 instructions,
// so none of them lags at the next barrier (a reduction left to ONE wave is time-sliced with the seven other
// waves of its SIMD and holds its workgroup back eight times its own length).
//   x4: the lane's elements; L: valid flat elements of this workgroup; inv_rows = 1 / (L / dim);
//   lds: 2 * fast_lds_floats() floats.
__host__ __device__ inline bool fast_flat_ok(int dim) { return dim >= 4 && dim <= 256 && (dim & (dim - 1)) == 0; }
__host__ __device__ inline int fast_lds_floats() { return 1024; }

// total of the wave in lane 63 (DPP only: the row totals, then row_bcast:15 / row_bcast:31 carry them upwards)
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
  auto dpp = [](float x, auto ctrl, auto rows) {
    return __int_as_float(
        __builtin_amdgcn_update_dpp(0, __float_as_int(x), decltype(ctrl)::value, decltype(rows)::value, 0xF, true));
  };
  using all = std::integral_constant<int, 0xF>;
  v += dpp(v, std::integral_constant<int, 0xB1>{}, all{});   // quad_perm [1,0,3,2]
  v += dpp(v, std::integral_constant<int, 0x4E>{}, all{});   // quad_perm [2,3,0,1]
  v += dpp(v, std::integral_constant<int, 0x141>{}, all{});  // row_half_mirror
  v += dpp(v, std::integral_constant<int, 0x140>{}, all{});  // row_mirror: every lane holds its row's total
  v += dpp(v, std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xA>{});  // row_bcast:15 into rows 1, 3
  v += dpp(v, std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xC>{});  // row_bcast:31 into rows 2, 3
  return v;
}

// inv_rows = 1 / (valid rows of this workgroup) (0 when it has none): the rows of a full workgroup are a power of two,
// where the product is the quotient; in the one ragged workgroup the centre is an ulp off the block mean, which the
// merge's pairwise identity does not notice (it shifts M2 by rows * ulp^2).
__device__ __forceinline__ void emit_flat_fast(const DiagArgs d, int keep, float* lds, int dim, float4 x4, int L, float inv_rows,
                                               float e_part) {
  // The lane geometry is recomputed here at every kept step (a dozen integer instructions): left to the compiler it
  // is hoisted out of the step loop, does not fit the 64 registers of eight waves per SIMD, and comes back from
  // scratch one dependent reload at a time right behind the barrier.
  int tid = threadIdx.x;
  asm volatile("" : "+v"(tid));
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  float* tile = lds + (keep & 1) * fast_lds_floats();
  *reinterpret_cast<float4*>(tile + 4 * tid) = x4;
  // ONE barrier: the tiles alternate with the kept step, and no wave can reach the barrier of the step after next
  // (behind which this tile is written again) before it has finished reading here
  __syncthreads();
  const int cw = dim >> 2;                 // columns per wave (1 .. 64)
  const int sh = __builtin_ctz(cw);        // 64 / cw lanes share a column; lane takes rows lane / cw + (64 / cw) i, i < 4
  const int col = wave * cw + (lane & (cw - 1));
  const int e0 = (lane >> sh) * dim + col;  // the four elements sit 256 apart: (64 / cw) rows = 256 floats
  float v[4];
  bool ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float t = tile[e0 + 256 * i];    // always inside the tile
    ok[i] = e0 + 256 * i < L;
    v[i] = ok[i] ? t : 0.0f;
  }
  const int lane4 = lane << 2;
  float sum = (v[0] + v[1]) + (v[2] + v[3]);
  for (int m4 = cw << 2; m4 < 256; m4 <<= 1)
    sum += __int_as_float(__builtin_amdgcn_ds_bpermute(lane4 ^ m4, __float_as_int(sum)));
  const float mu = sum * inv_rows;
  float m2 = 0.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float dv = ok[i] ? v[i] - mu : 0.0f;
    m2 = __builtin_fmaf(dv, dv, m2);
  }
  for (int m4 = cw << 2; m4 < 256; m4 <<= 1)
    m2 += __int_as_float(__builtin_amdgcn_ds_bpermute(lane4 ^ m4, __float_as_int(m2)));
  float* rec = d.partials + ((int64_t)keep * d.n_blocks + blockIdx.x) * (int64_t)record_floats(d.S);
  if (lane < cw) {
    rec[col] = sum;
    rec[d.S + col] = m2;
  }
  const float e = wave_sum_to_lane63(e_part);
  if (lane == 63) {  // tail: four per-wave energy shares, four accept shares
    rec[2 * d.S + wave] = e;
    rec[2 * d.S + 4 + wave] = 0.0f;
  }
}

// ------------------------------------------------------------------------------------------------------------------
// Records of the MATRIX-LAYOUT kernels (dense Gaussian / mixture / MLP on the matrix cores).  The state of a wave's 32 chains
// sits in the 32x32 C/D layout -- lane (m, h), register r of tile t = coordinate 32 t + (r & 3) + 8 (r >> 2) + 4 h of chain m
// -- so a column statistic is a sum over the 32 lanes of a K-half: no LDS tile, no barrier.  A WAVE is a "block" of the record
// geometry: E = 32 dim flat elements, S = dim slots, n_blocks = ceil(n / 32); diag_finish_kernel merges them like any other
// layout.  Two passes (sum -> wave mean -> squared deviations), as in emit(): no cancellation.
// ------------------------------------------------------------------------------------------------------------------
// Sum over the 32 lanes that share this lane's h, in every lane.  Four DPP adds inside the row of 16 (quad swaps, half-row
// mirror, row mirror: no LDS traffic, no wait) and ONE ds_swizzle for the other row (lane ^ 16).  As five __shfl_xor it was five
// dependent ds_bpermute round trips per value -- 10 per record column, ~14 k cycles per record of a 32-wide state: 60 % of an
// MLP Langevin step (scripts/bench_mlp_diag_kernel.py).
template <int CTRL>
__device__ __forceinline__ float dpp_get(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_wave_sum(float v) {
  v += dpp_get<0xB1>(v);   // quad_perm [1, 0, 3, 2]
  v += dpp_get<0x4E>(v);   // quad_perm [2, 3, 0, 1]
  v += dpp_get<0x141>(v);  // row_half_mirror: the other quad of the eight
  v += dpp_get<0x140>(v);  // row_mirror: the other eight of the sixteen
  v += __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));  // bit mode: and 0x1f, xor 0x10
  return v;
}

// sums and centred second moments of the wave's chains at kept step `keep`; at(t, r): register r of tile t of this lane.
// The energy / accept shares follow through wave_record_tail() (the MLP Langevin kernel: one evaluation later).
template <int NT, class At>
__device__ __forceinline__ void wave_record(float* partials, int64_t n_blocks, int keep, int64_t wave_id, int dim, At at, bool active,
                                            int lane, int tile0 = 0, int col0 = 0) {  // tile0: the first tile's index in the row (a slice of it)
  // col0: tile coordinate of the row's column 0 (shifted rows, gauss_mfma_body.h SH; tile coordinates outside the row: no column)
  if (wave_id >= n_blocks) return;  // a wave past the last chain has no record (wave-uniform)
  const int h = lane >> 5;
  const int valid = __popcll(__ballot(active)) >> 1;  // both K-halves of a chain vote
  const float inv = valid > 0 ? 1.0f / (float)valid : 0.0f;
  float* rec = partials + ((int64_t)keep * n_blocks + wave_id) * (int64_t)record_floats(dim);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float v = active ? at(t, r) : 0.0f;
      const float sum = half_wave_sum(v);
      const float dv = active ? v - sum * inv : 0.0f;
      const float m2 = half_wave_sum(dv * dv);
      const int c = 32 * (t + tile0) + (r & 3) + 8 * (r >> 2) + 4 * h - col0;
      if ((lane & 31) == 0 && c >= 0 && c < dim) {
        rec[c] = sum;
        rec[dim + c] = m2;
      }
    }
}
__device__ __forceinline__ void wave_record_tail(float* partials, int64_t n_blocks, int keep, int64_t wave_id, int dim, float energy,
                                                 bool active, bool accepted, int lane) {
  if (wave_id >= n_blocks) return;
  float e = (active && lane < 32) ? energy : 0.0f;  // one K-half speaks for the chain
  e = wave_sum(e);
  const int acc = __popcll(__ballot(accepted && lane < 32));
  if (lane < 8) {
    float* rec = partials + ((int64_t)keep * n_blocks + wave_id) * (int64_t)record_floats(dim);
    rec[2 * dim + lane] = lane == 0 ? e : (lane == 4 ? (float)acc : 0.0f);
  }
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

// Records of INTERLEAVED CLASSES (shifted rows): record b = (group b / K, class b % K) holds chains 32 K g + K m + s,
// m = 0 .. 31 -- K waves of different workgroups share a run of 32 K consecutive chains.  Signalled to the caller and to
// ebm_diag_finish_f32 by a NEGATIVE block_elems (-32 dim); K = 4 / gcd(dim, 4).
inline int diag_classes(int32_t dim) { return (dim & 1) ? 4 : ((dim & 2) ? 2 : 1); }
inline bool plan_classes(int64_t n_chains, int32_t dim, DiagArgs& d) {
  const int K = diag_classes(dim);
  d.E = -32 * dim;
  d.S = dim;
  d.n_blocks = ceil_div64(n_chains, 32 * (int64_t)K) * K;
  return true;
}

}  // namespace diag
}  // namespace ebm
