// Small streaming kernels around the hot path: native-RNG fill, the leapfrog sub-steps and
// Metropolis accept used when the force comes from an opaque drift closure (autograd
// models), and the column statistics behind the sampler diagnostics.
#include "diag.h"
#include "ebm_common.h"

namespace ebm {
namespace {

constexpr int kBlock = 256;
constexpr int kMaxGrid = 256 * 8;  // 256 CUs x 8 blocks; grid-stride beyond that

int grid_for(int64_t n_threads) {
  int64_t b = ceil_div64(n_threads, kBlock);
  if (b < 1) b = 1;
  if (b > kMaxGrid) b = kMaxGrid;
  return (int)b;
}

// ---------------------------------------------------------------------------------
// rng_dev (optional): {seed, step} in device memory; the launch's own `step` is then an offset from
// it -- what lets a launch captured in a HIP graph draw fresh numbers on every replay.
__device__ __forceinline__ void resolve_rng(const uint64_t* rng_dev, RngKey& key, uint64_t& step) {
  if (rng_dev) {
    const uint64_t seed = rng_dev[0];
    key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
    step += rng_dev[1];
  }
}

__global__ __launch_bounds__(kBlock) void noise_fill_kernel(float* __restrict__ out, int64_t n_elem,
                                                            int kind, RngKey key, uint64_t step,
                                                            const uint64_t* __restrict__ rng_dev) {
  resolve_rng(rng_dev, key, step);
  const int64_t n_groups = ceil_div64(n_elem, 4);
  for (int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x; g < n_groups;
       g += (int64_t)gridDim.x * kBlock) {
    float r[4];
    if (kind == EBM_NOISE_NORMAL) {
      const F4 n = normal4_at(key, (uint64_t)g, step);
#pragma unroll
      for (int i = 0; i < 4; ++i) r[i] = n.v[i];
    } else {
      const U4 o = philox_at(key, (uint64_t)g, step);
      const uint32_t w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
        r[i] = (kind == EBM_NOISE_UNIFORM) ? u01_half_open(w[i]) : __uint_as_float(w[i]);
    }
    const int64_t e0 = g * 4;
    if (e0 + 4 <= n_elem) {
      // (a non-temporal store was measured SLOWER for this write-only stream: 0.064 vs 0.053 ms at 2^26)
      *reinterpret_cast<float4*>(out + e0) = make_float4(r[0], r[1], r[2], r[3]);
    } else {
      for (int i = 0; i < 4; ++i)
        if (e0 + i < n_elem) out[e0 + i] = r[i];
    }
  }
}

// ---------------------------------------------------------------------------------
// leapfrog sub-steps (integrators/leapfrog.py:156-185), arithmetic in reference order:
//   (0.5*eps) is formed first as an fp32 scalar tensor, then multiplied into the force.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void kick_drift_kernel(
    const float* __restrict__ x, const float* __restrict__ p, const float* __restrict__ force,
    float* __restrict__ x_new, float* __restrict__ p_half, int64_t n_elem, int32_t dim, float eps,
    float half_eps, int mass_kind, float mass_scalar, const float* __restrict__ mass_diag, int safe) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_elem;
       e += (int64_t)gridDim.x * kBlock) {
    float f = force[e];
    if (safe) f = clamp_nanprop(f, -1e6f, 1e6f);
    const float ph = p[e] + half_eps * f;
    float xn;
    if (mass_kind == EBM_MASS_NONE) {
      xn = x[e] + eps * ph;
    } else {
      float m = mass_scalar;  // already max(mass, 1e-10) for the scalar form
      if (mass_kind == EBM_MASS_DIAG) {
        m = mass_diag[e % dim];
        m = (m < 1e-10f) ? 1e-10f : m;  // torch.clamp(mass, min=1e-10)
      }
      xn = x[e] + (eps * ph) / m;
    }
    p_half[e] = ph;
    x_new[e] = xn;
  }
}

__global__ __launch_bounds__(kBlock) void kick_kernel(float* __restrict__ x_new,
                                                      const float* __restrict__ p_half,
                                                      const float* __restrict__ force,
                                                      float* __restrict__ p_new, int64_t n_elem,
                                                      float half_eps, int safe) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_elem;
       e += (int64_t)gridDim.x * kBlock) {
    float f = force[e];
    if (safe) f = clamp_nanprop(f, -1e6f, 1e6f);
    float pn = p_half[e] + half_eps * f;
    if (safe) {
      pn = nan_to_num0(pn);
      const float xv = x_new[e];
      const float xs = nan_to_num0(xv);
      if (!(xs == xv)) x_new[e] = xs;  // store only when the scrub changed something
    }
    p_new[e] = pn;
  }
}

// ---------------------------------------------------------------------------------
// noise-free descent updates with an external gradient (samplers/gradient_descent.py:121-123,
// :262-266): torch.sub/add with alpha are FMAs, and so are these.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void descent_step_kernel(const float* __restrict__ x,
                                                              const float* __restrict__ grad,
                                                              float* v, float* out, int64_t n_elem,
                                                              float neg_eta, float mu) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_elem;
       e += (int64_t)gridDim.x * kBlock) {
    if (v) {
      const float vn = __builtin_fmaf(neg_eta, grad[e], v[e] * mu);
      v[e] = vn;
      out[e] = x[e] + vn;
    } else {
      out[e] = __builtin_fmaf(neg_eta, grad[e], x[e]);
    }
  }
}

__global__ __launch_bounds__(kBlock) void lookahead_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ v,
                                                           float* __restrict__ out, int64_t n_elem, float mu) {
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n_elem;
       e += (int64_t)gridDim.x * kBlock)
    out[e] = __builtin_fmaf(mu, v[e], x[e]);
}

// ---------------------------------------------------------------------------------
// persistent-CD replay buffer: stratified gather and FIFO scatter, one element per lane so that a
// row of `dim` floats is read/written by consecutive lanes (core/base_loss.py:296-315, :390-426)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void pcd_gather_kernel(const float* __restrict__ buffer, int64_t buffer_size,
                                                            int32_t dim, float* __restrict__ out, int64_t batch,
                                                            int64_t stride, const int64_t* __restrict__ offsets,
                                                            int64_t* __restrict__ rows_out, RngKey key, uint64_t step,
                                                            const uint64_t* __restrict__ rng_dev) {
  resolve_rng(rng_dev, key, step);
  const int64_t n = batch * dim;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
    const int64_t i = e / dim;
    const int d = (int)(e - i * dim);
    int64_t r;
    if (offsets) {
      r = offsets[i];
    } else {
      const uint32_t o = pick(philox_at(key, (uint64_t)i >> 2, step), (int)(i & 3));
      r = (int64_t)(((uint64_t)o * (uint64_t)stride) >> 32);  // multiply-shift: uniform in [0, stride)
    }
    const int64_t row = (i * stride + r) % buffer_size;
    out[e] = buffer[row * dim + d];
    if (rows_out && d == 0) rows_out[i] = row;
  }
}

// ... and the same gather with the reference's exploration noise folded in (core/base_loss.py:316-332: `randperm(batch)[:n_new]` rows
// get `+ 0.01 randn`): the random subset is { i : pi(i) < n_noise } for a keyed bijection pi of [0, batch) -- a six-round Feistel
// network on ceil(log2 batch) bits (alternating halves, a 32-bit multiply-xorshift mixer as round function, six round keys from the
// launch's Philox field) walked until it lands inside [0, batch) -- so exactly n_noise rows are chosen, each subset of that size
// (pseudo-)equally likely, and no sort, no index list and no second pass over the rows is needed.  Steps of the field: `step` the
// in-stride offsets (as ebm_pcd_gather_f32), `step + 1` the round keys (groups 0 and 1), `step + 2` the normals (element e of out).
__device__ __forceinline__ uint32_t mix32(uint32_t h) {
  h ^= h >> 16; h *= 0x21f0aaadu; h ^= h >> 15; h *= 0x735a2d97u; h ^= h >> 15;
  return h;
}
__device__ __forceinline__ uint32_t feistel6(uint32_t v, int a_bits, int b_bits, const uint32_t (&k)[6]) {
  uint32_t L = v >> b_bits, R = v & ((1u << b_bits) - 1u);
#pragma unroll
  for (int r = 0; r < 6; ++r) {
    if ((r & 1) == 0) L ^= mix32(R ^ k[r]) >> (32 - a_bits);
    else R ^= mix32(L ^ k[r]) >> (32 - b_bits);
  }
  return (L << b_bits) | R;
}

__global__ __launch_bounds__(kBlock) void pcd_start_points_kernel(const float* __restrict__ buffer, int64_t buffer_size, int32_t dim,
                                                                  float* __restrict__ out, int64_t batch, int64_t stride,
                                                                  int64_t n_noise, float noise_scale, int bits, RngKey key, uint64_t step,
                                                                  const uint64_t* __restrict__ rng_dev) {
  resolve_rng(rng_dev, key, step);
  const U4 k0 = philox_at(key, 0, step + 1), k1 = philox_at(key, 1, step + 1);
  const uint32_t rk[6] = {k0.x, k0.y, k0.z, k0.w, k1.x, k1.y};
  const int a_bits = bits >> 1, b_bits = bits - a_bits;
  const int64_t n = batch * dim;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
    const int64_t i = e / dim;
    const int d = (int)(e - i * dim);
    const uint32_t o = pick(philox_at(key, (uint64_t)i >> 2, step), (int)(i & 3));
    const int64_t r = (int64_t)(((uint64_t)o * (uint64_t)stride) >> 32);  // multiply-shift: uniform in [0, stride)
    const int64_t row = (i * stride + r) % buffer_size;
    float v = buffer[row * dim + d];
    uint32_t p = (uint32_t)i;
    do p = feistel6(p, a_bits, b_bits, rk); while ((int64_t)p >= batch);  // cycle walking: a bijection of [0, batch)
    if ((int64_t)p < n_noise) {
      const F4 z = normal4_at(key, (uint64_t)e >> 2, step + 2);
      const int c = (int)(e & 3);
      const float zn = c == 0 ? z.v[0] : c == 1 ? z.v[1] : c == 2 ? z.v[2] : z.v[3];
      v = v + zn * noise_scale;
    }
    out[e] = v;
  }
}

__global__ __launch_bounds__(kBlock) void pcd_scatter_kernel(float* __restrict__ buffer, int64_t buffer_size,
                                                             int32_t dim, const float* __restrict__ samples,
                                                             int64_t batch, int64_t write_pos,
                                                             const int64_t* __restrict__ pos_dev) {
  if (pos_dev) write_pos = ((*pos_dev % buffer_size) + buffer_size) % buffer_size;  // (a position out of range cannot leave the buffer)
  const int64_t n = batch * dim;
  for (int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x; e < n; e += (int64_t)gridDim.x * kBlock) {
    const int64_t i = e / dim;
    const int64_t row = (write_pos + i) % buffer_size;
    buffer[row * dim + (e - i * dim)] = samples[e];
  }
}

// ---------------------------------------------------------------------------------
// The contrastive-divergence loss of one model call on [data | negatives] (losses/contrastive_divergence.py:128-155):
//   L = mean(E+) - mean(E-) + reg (mean(E+^2) + mean(E-^2)),  a non-finite L becomes the constant 0.1 and sends no gradient.
// Forward: one launch -- block partials of the four sums in fp64, the last block to finish adds them in block order (a ticket
// counter it leaves at zero again) and writes L and the finite flag: the value depends on no scheduling.  Backward: one
// elementwise launch, dL/dE_i = (+-1 + 2 reg E_i) / n * (upstream * finite).  torch's graph of the same arithmetic is some
// twenty launches of 4 - 5 us on two scalars and two vectors -- a tenth of a captured training step.
// ---------------------------------------------------------------------------------
constexpr int kCdLossBlocks = 128;
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) v += __shfl_down(v, m, 64);
  return v;
}
__global__ __launch_bounds__(kBlock) void cd_loss_kernel(const float* __restrict__ e_both, int64_t n, float reg, double* __restrict__ partials,
                                                         unsigned int* __restrict__ ticket, float* __restrict__ loss_out,
                                                         float* __restrict__ finite_out) {
  double s[4] = {0.0, 0.0, 0.0, 0.0};  // sum E+, sum E-, sum E+^2, sum E-^2
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) {
    const double p = e_both[i], q = e_both[n + i];
    s[0] += p; s[1] += q; s[2] += p * p; s[3] += q * q;
  }
  __shared__ double part[kBlock / 64][4];
  __shared__ bool last;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const double w = wave_sum(s[c]);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][c] = w;
  }
  __syncthreads();
  if (threadIdx.x < 4) {
    double b = 0.0;
    for (int w = 0; w < kBlock / 64; ++w) b += part[w][threadIdx.x];
    partials[(int64_t)blockIdx.x * 4 + threadIdx.x] = b;
    __threadfence();
  }
  __syncthreads();
  if (threadIdx.x == 0) last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
  __syncthreads();
  if (!last) return;
  __threadfence();
  {  // wave c adds column c of the partials: lane l takes blocks l, l + 64, ... in order, then the fixed shuffle tree
    const int c = threadIdx.x >> 6, l = threadIdx.x & 63;
    double t = 0.0;
    for (unsigned g = l; g < gridDim.x; g += 64) t += __builtin_nontemporal_load(&partials[(int64_t)g * 4 + c]);
    t = wave_sum(t);
    __syncthreads();  // (part[] of the block's own partial sums has been read)
    if (l == 0) part[0][c] = t / (double)n;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // the reference's order of operations on fp32 means: (mean+ - mean-) + reg * (meansq+ + meansq-)
    const float mp = (float)part[0][0], mn = (float)part[0][1], sp = (float)part[0][2], sn = (float)part[0][3];
    float loss = mp - mn;
    if (reg > 0.0f) loss = loss + reg * (sp + sn);
    const bool ok = isfinite(loss);
    *loss_out = ok ? loss : 0.1f;
    *finite_out = ok ? 1.0f : 0.0f;
    *ticket = 0u;
  }
}

__global__ __launch_bounds__(kBlock) void cd_loss_seed_kernel(const float* __restrict__ e_both, int64_t n, float reg,
                                                              const float* __restrict__ upstream, const float* __restrict__ finite,
                                                              float* __restrict__ seed_out) {
  const float scale = *upstream * *finite;  // (a non-finite loss sends no gradient: contrastive_divergence.py:150-155)
  const float inv_n = 1.0f / (float)n, two_reg_n = 2.0f * reg / (float)n;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < 2 * n; i += (int64_t)gridDim.x * kBlock) {
    float g = i < n ? inv_n : -inv_n;
    if (reg > 0.0f) g = g + e_both[i] * two_reg_n;
    seed_out[i] = g * scale;
  }
}

// ---------------------------------------------------------------------------------
// Metropolis accept, one lane-group of `lanes` lanes per chain row (samplers/hmc.py:277-292)
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void hmc_accept_kernel(
    float* __restrict__ x, const float* __restrict__ x_prop, const float* __restrict__ h0,
    const float* __restrict__ h1, const float* __restrict__ u, uint8_t* __restrict__ mask,
    uint32_t* __restrict__ count, int64_t n_chains, int32_t dim, RngKey key, uint64_t step,
    const uint64_t* __restrict__ rng_dev) {
  resolve_rng(rng_dev, key, step);
  // phase 1: one lane per chain decides; phase 2: the block copies accepted rows.
  __shared__ uint8_t acc_s[kBlock];
  for (int64_t c0 = (int64_t)blockIdx.x * kBlock; c0 < n_chains; c0 += (int64_t)gridDim.x * kBlock) {
    const int64_t c = c0 + threadIdx.x;
    bool acc = false;
    if (c < n_chains) {
      const float d = clamp_nanprop(h0[c] - h1[c], -50.0f, 50.0f);
      float a = expf(d);
      a = (a > 1.0f) ? 1.0f : a;  // clamp_(max=1), NaN stays NaN
      float uu;
      if (u) uu = u[c];
      else uu = u01_half_open(pick(philox_at(key, (uint64_t)(c >> 2), step), (int)(c & 3)));
      acc = uu < a;
      if (mask) mask[c] = acc ? 1 : 0;
    }
    if (count) {
      const unsigned long long b = __ballot(acc);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, (uint32_t)__popcll(b));
    }
    acc_s[threadIdx.x] = acc ? 1 : 0;
    __syncthreads();
    const int64_t rows = (n_chains - c0) < kBlock ? (n_chains - c0) : kBlock;
    const int64_t n = rows * dim;
    for (int64_t i = threadIdx.x; i < n; i += kBlock) {
      const int64_t r = i / dim;
      if (acc_s[r]) x[c0 * dim + i] = x_prop[c0 * dim + i];
    }
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------
// column statistics: one pass with shifted fp64 accumulators, per-block LDS partials, then one fp64
// atomic per column per block.
// ---------------------------------------------------------------------------------
typedef float v4f_stat __attribute__((ext_vector_type(4)));  // the vector form the non-temporal builtin accepts
constexpr int kStatCols = 64;   // columns per block tile (== wave size: see the ticket logic)
constexpr int kStatRows = 4;    // row lanes per block tile (kBlock / kStatCols)

// One pass, shifted sums in fp64: with s_c = x[0, c] as the shift, S1 = sum (x - s), S2 = sum (x - s)^2;
// mean = s + S1/n, var = (S2 - S1^2/n)/n.  The shift removes the cancellation of the textbook
// one-pass formula when |mean| >> std, fp64 accumulators remove the rest.
__global__ __launch_bounds__(kBlock) void chain_stats_kernel(const float* __restrict__ x,
                                                             int64_t n_chains, int32_t dim,
                                                             double* __restrict__ work,
                                                             float* __restrict__ mean_out,
                                                             float* __restrict__ var_out) {
  __shared__ double part1[kStatRows][kStatCols];
  __shared__ double part2[kStatRows][kStatCols];
  const int col = blockIdx.x * kStatCols + (threadIdx.x % kStatCols);
  const int rlane = threadIdx.x / kStatCols;
  double s1 = 0.0, s2 = 0.0;
  if (col < dim) {
    const double shift = (double)x[col];
#pragma unroll 8  // eight independent row loads in flight per lane
    for (int64_t r = (int64_t)blockIdx.y * kStatRows + rlane; r < n_chains;
         r += (int64_t)gridDim.y * kStatRows) {
      const double v = (double)x[r * dim + col] - shift;
      s1 += v;
      s2 += v * v;
    }
  }
  part1[rlane][threadIdx.x % kStatCols] = s1;
  part2[rlane][threadIdx.x % kStatCols] = s2;
  __syncthreads();
  if (rlane == 0 && col < dim) {
    double a1 = 0.0, a2 = 0.0;
#pragma unroll
    for (int i = 0; i < kStatRows; ++i) {
      a1 += part1[i][threadIdx.x];
      a2 += part2[i][threadIdx.x];
    }
    // returning atomics: once the returned values are here the adds have been performed at L2
    const double r1 = atomicAdd(&work[col], a1);
    const double r2 = atomicAdd(&work[dim + col], a2);
    asm volatile("" ::"v"(r1), "v"(r2));
  }
  // The last block to arrive turns the sums into mean / var and leaves `work` zeroed for the next call:
  // one launch per diagnostics step, no memset and no finishing kernel.  All of a block's atomics and its
  // ticket are issued by wave 0 (kStatCols == 64) in program order, the sums are read back with
  // atomics at L2, so no device-wide fence (an L2 write-back on this chip) is needed.
  __shared__ bool last;
  if (threadIdx.x == 0) {
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(work + 2 * (int64_t)dim);
    const unsigned long long total = (unsigned long long)gridDim.x * gridDim.y;
    last = atomicAdd(ticket, 1ull) == total - 1ull;
  }
  __syncthreads();
  if (!last) return;
  const double n = (double)n_chains;
  for (int c = threadIdx.x; c < dim; c += kBlock) {
    const double s1 = atomicExch(&work[c], 0.0);          // read at L2 (where the atomics landed) and reset
    const double s2 = atomicExch(&work[dim + c], 0.0);
    mean_out[c] = (float)((double)x[c] + s1 / n);
    float v = (float)((s2 - s1 * s1 / n) / n);
    var_out[c] = clamp_nanprop(v, 1e-10f, 1e10f);  // langevin_dynamics.py:176-178
  }
  if (threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(work + 2 * (int64_t)dim) = 0ull;
}

// Fast path for dim a power of two in [4, 1024]: the matrix is walked as a flat float4 stream (a wave
// reads 1 KiB contiguous, like the update kernels), a lane's four columns are fixed because a block tile
// of 1024 elements is a whole number of rows, per-lane fp64 sums, LDS fp64 atomics across the lanes that
// share columns, then the same global atomics + last-block finish as above.
__global__ __launch_bounds__(kBlock) void chain_stats_wide_kernel(const float* __restrict__ x, int64_t n_chains,
                                                                  int32_t dim, double* __restrict__ work,
                                                                  float* __restrict__ mean_out,
                                                                  float* __restrict__ var_out) {
  __shared__ double acc1[1024];
  __shared__ double acc2[1024];
  for (int c = threadIdx.x; c < dim; c += kBlock) acc1[c] = acc2[c] = 0.0;
  __syncthreads();
  const int64_t n_groups = n_chains * (int64_t)dim / 4;
  const int col0 = (threadIdx.x * 4) & (dim - 1);
  const float4 sh4 = *reinterpret_cast<const float4*>(x + col0);  // row 0 of the lane's columns: the shift
  const v4f_stat sh = {sh4.x, sh4.y, sh4.z, sh4.w};
  double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
  // four rows at a time in fp32 (eight measured slower) (shifted values, so the short sums lose nothing that matters), then one
  // fold into the fp64 accumulators: an eighth of the fp64 work of converting every element
  const int64_t stride = (int64_t)gridDim.x * kBlock;
  int64_t g = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  for (; g + 3 * stride < n_groups; g += 4 * stride) {
    v4f_stat t1 = {0.f, 0.f, 0.f, 0.f}, t2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const v4f_stat d = __builtin_nontemporal_load(reinterpret_cast<const v4f_stat*>(x) + g + j * stride) - sh;
      t1 += d;
      t2 = __builtin_elementwise_fma(d, d, t2);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s1[i] += (double)t1[i];
      s2[i] += (double)t2[i];
    }
  }
  for (; g < n_groups; g += stride) {
    const v4f_stat d = __builtin_nontemporal_load(reinterpret_cast<const v4f_stat*>(x) + g) - sh;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s1[i] += (double)d[i];
      s2[i] += (double)d[i] * (double)d[i];
    }
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    atomicAdd(&acc1[col0 + i], s1[i]);
    atomicAdd(&acc2[col0 + i], s2[i]);
  }
  __syncthreads();
  for (int c = threadIdx.x; c < dim; c += kBlock) {
    const double r1 = atomicAdd(&work[c], acc1[c]);
    const double r2 = atomicAdd(&work[dim + c], acc2[c]);
    asm volatile("" ::"v"(r1), "v"(r2));
  }
  __shared__ bool last;
  __syncthreads();  // every wave's returning atomics have completed before the ticket is taken
  if (threadIdx.x == 0) {
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(work + 2 * (int64_t)dim);
    last = atomicAdd(ticket, 1ull) == (unsigned long long)gridDim.x - 1ull;
  }
  __syncthreads();
  if (!last) return;
  const double n = (double)n_chains;
  for (int c = threadIdx.x; c < dim; c += kBlock) {
    const double t1 = atomicExch(&work[c], 0.0);
    const double t2 = atomicExch(&work[dim + c], 0.0);
    mean_out[c] = (float)((double)x[c] + t1 / n);
    float v = (float)((t2 - t1 * t1 / n) / n);
    var_out[c] = clamp_nanprop(v, 1e-10f, 1e10f);
  }
  if (threadIdx.x == 0) *reinterpret_cast<unsigned long long*>(work + 2 * (int64_t)dim) = 0ull;
}


// ---------------------------------------------------------------------------------
// Calibration probe (bench.py): a dependent-free stream of plain full-rate VALU ops (v_fma_f32 on eight
// independent registers per lane).  Its duration gives the chip's plain-VALU issue rate on THIS box at
// THIS clock, which is what the VALU-bound chain kernels are priced against.
// ---------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void probe_valu_kernel(float* __restrict__ out, int iters) {
  float a[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 1e-3f + j;
  float m = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f;
  asm volatile("" : "+v"(m), "+v"(c));
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(m), "v"(c));  // (literal: left to the
    // compiler, adjacent FMAs are SLP-packed into v_pk_fma_f32 -- the same issue time per FMA on this part, but not what the name says)
  }
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) s += a[j];
  out[(int64_t)blockIdx.x * kBlock + threadIdx.x] = s;
}


// ---------------------------------------------------------------------------------
// Merge of the per-block diagnostics records the chain kernels emit (diag.h) into the sampler
// diagnostics: mean[n_kept, dim], biased var[n_kept, dim] clamped to [1e-10, 1e10] (0 for a single chain),
// energy[n_kept] = mean per-chain energy, accept[n_kept] = accepted fraction.
//
// grid = (n_kept, W, Q): W = column windows (1 when a block holds whole rows, dim / E when a row spans several
// blocks: block b then covers columns (b % W) * E + slot), Q splits the block range.  Every workgroup adds its
// share of three plain sums per column to the fp64 work row of its kept step --
//   A = sum_b sum_x,   B = sum_b M2_b,   C = sum_b cnt_b (mean_b - shift)^2,   shift = mean of the window's first block
// -- and M2_total = B + C - n (mean - shift)^2 (pairwise-variance identity about a common shift; exact in
// exact arithmetic, and in fp64 conditioned by |mean_b - shift| ~ the spread of block means, not by |mean|).
// The last workgroup of a kept step (ticket) turns the sums into the outputs and leaves the work row zeroed.
// work: double[n_kept][3 * dim + 3] = {A[dim], B[dim], C[dim], energy sum, accept sum, ticket}, zeroed by the caller.
// ---------------------------------------------------------------------------------
// K > 1: records of K interleaved classes (diag.h plan_classes): record b holds the chains 32 K (b / K) + K m + (b % K)
__device__ __forceinline__ int64_t diag_block_len(int64_t b, int E, int64_t n_elem, int K = 1, int dim = 1) {
  if (K > 1) {
    const int64_t n = n_elem / dim, first = 32 * (int64_t)K * (b / K) + (b % K);
    const int64_t rows = first < n ? (n - first + K - 1) / K : 0;
    return (rows > 32 ? 32 : rows) * dim;
  }
  const int64_t left = n_elem - b * (int64_t)E;
  return left >= E ? E : (left > 0 ? left : 0);
}

__global__ __launch_bounds__(kBlock) void diag_finish_kernel(const float* __restrict__ partials, int64_t n_blocks,
                                                             int32_t S, int32_t E, int64_t n_chains, int32_t dim,
                                                             float* __restrict__ mean_out, float* __restrict__ var_out,
                                                             float* __restrict__ energy_out, float* __restrict__ accept_out,
                                                             double* __restrict__ work, int K) {
  const int j = blockIdx.x;
  const int W = gridDim.y, w = blockIdx.y, Q = gridDim.z, q = blockIdx.z;
  const int R = diag::record_floats(S);
  const int64_t n_elem = n_chains * (int64_t)dim;
  const float* base = partials + (int64_t)j * n_blocks * R;
  double* wrow = work + (int64_t)j * (3 * (int64_t)dim + 3);
  // blocks of this window: b = w, w + W, ...; this workgroup takes the q-th chunk of them
  const int64_t n_win = (n_blocks - w + W - 1) / W;
  const int64_t per = (n_win + Q - 1) / Q;
  const int64_t i0 = (int64_t)q * per, i1 = (i0 + per < n_win) ? i0 + per : n_win;
  const int64_t len_first = diag_block_len(w, E, n_elem, K, dim);
  // lanes: slot s = t % SP, record lane p = t / SP of P (small S: several lanes walk the same slot over
  // interleaved records); their partial sums meet in LDS before the atomics
  __shared__ double red[3][kBlock];
  const int SP = S < kBlock ? S : kBlock;
  const int P = kBlock / SP;
  const int p = threadIdx.x / SP;
  const double inv_full = 1.0 / (double)((E + dim - 1) / dim);  // 1 / rows of a full block's slot (E % dim == 0), or 1
  for (int s0 = 0; s0 < S; s0 += SP) {
    const int s = s0 + (threadIdx.x - p * SP);
    double A = 0.0, B = 0.0, C = 0.0;
    double shift = 0.0;
    if (p < P && s < S) {
      const int cnt0 = s < len_first ? (int)((len_first - s + dim - 1) / dim) : 0;
      shift = cnt0 > 0 ? (double)base[(int64_t)w * R + s] / (double)cnt0 : 0.0;
      // full blocks first, four records in flight per lane (one record per trip leaves the loop waiting on a single
      // pair of loads); the ragged tail of the state goes through the general trip below
      const int64_t n_full = K > 1 ? (n_chains / (32 * (int64_t)K)) * K : n_elem / E;
      const double cnt_full = (double)((E + dim - 1) / dim);
      double Cf = 0.0;
      int64_t i = i0 + p;
      for (; i + 3 * P < i1 && w + (i + 3 * P) * W < n_full; i += 4 * P) {
        float sx[4], m2[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* rec = base + (w + (i + (int64_t)u * P) * W) * R;
          sx[u] = __builtin_nontemporal_load(rec + s);
          m2[u] = __builtin_nontemporal_load(rec + S + s);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const double dm = (double)sx[u] * inv_full - shift;
          A += (double)sx[u];
          B += (double)m2[u];
          Cf = __builtin_fma(dm, dm, Cf);
        }
      }
      C = cnt_full * Cf;
      for (; i < i1; i += P) {
        const int64_t b = w + i * W;
        const int64_t len = diag_block_len(b, E, n_elem, K, dim);
        if (s >= len) continue;
        const float* rec = base + b * R;
        const double sx = (double)rec[s];
        double cnt, inv;
        if (len == E) {
          cnt = (double)((E + dim - 1) / dim);
          inv = inv_full;
        } else {
          cnt = (double)((len - s + dim - 1) / dim);
          inv = 1.0 / cnt;
        }
        const double dm = sx * inv - shift;
        A += sx;
        B += (double)rec[S + s];
        C += cnt * dm * dm;
      }
    }
    red[0][threadIdx.x] = A; red[1][threadIdx.x] = B; red[2][threadIdx.x] = C;
    __syncthreads();
    if (p == 0 && s < S) {
      for (int k = 1; k < P; ++k) {
        A += red[0][threadIdx.x + k * SP]; B += red[1][threadIdx.x + k * SP]; C += red[2][threadIdx.x + k * SP];
      }
      const int col = w * E + s;  // W == 1: col = s
      const double r0 = atomicAdd(&wrow[col], A);
      const double r1 = atomicAdd(&wrow[dim + col], B);
      const double r2 = atomicAdd(&wrow[2 * dim + col], C);
      asm volatile("" ::"v"(r0), "v"(r1), "v"(r2));  // the adds have been performed at L2 once their old values are back
    }
    __syncthreads();
  }
  {  // energy / accept sums of this workgroup's blocks: a record per lane and trip (ONE lane walking them all was
     // the whole run time of this kernel: 6.7 ms for 200 kept steps of config 2), then one total per workgroup
    double es = 0.0, as = 0.0;
    for (int64_t i = i0 + threadIdx.x; i < i1; i += kBlock) {
      const float* tail = base + (w + i * W) * R + 2 * S;
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = __builtin_nontemporal_load(tail + u);
      es += ((double)t[0] + (double)t[1]) + ((double)t[2] + (double)t[3]);
      as += ((double)t[4] + (double)t[5]) + ((double)t[6] + (double)t[7]);
    }
    red[0][threadIdx.x] = es; red[1][threadIdx.x] = as;
    __syncthreads();
    for (int h = kBlock / 2; h >= 1; h >>= 1) {
      if ((int)threadIdx.x < h) {
        red[0][threadIdx.x] += red[0][threadIdx.x + h];
        red[1][threadIdx.x] += red[1][threadIdx.x + h];
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      const double r0 = atomicAdd(&wrow[3 * dim], red[0][0]);
      const double r1 = atomicAdd(&wrow[3 * dim + 1], red[1][0]);
      asm volatile("" ::"v"(r0), "v"(r1));
    }
  }
  __shared__ bool last;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* ticket = reinterpret_cast<unsigned long long*>(wrow + 3 * dim + 2);
    last = atomicAdd(ticket, 1ull) == (unsigned long long)W * Q - 1ull;
  }
  __syncthreads();
  if (!last) return;
  const double n = (double)n_chains;
  for (int c = threadIdx.x; c < dim; c += kBlock) {
    const double A = atomicExch(&wrow[c], 0.0);  // read at L2 (where the atomics landed) and reset
    const double B = atomicExch(&wrow[dim + c], 0.0);
    const double C = atomicExch(&wrow[2 * dim + c], 0.0);
    const int wc = c / E, sc = c - wc * E;       // the shift this column's sums were taken about
    const int64_t lf = diag_block_len(wc, E, n_elem, K, dim);
    const int cnt0 = sc < lf ? (int)((lf - sc + dim - 1) / dim) : 0;
    const double shift = cnt0 > 0 ? (double)base[(int64_t)wc * R + sc] / (double)cnt0 : 0.0;
    const double mean = A / n;
    const double dm = mean - shift;
    double m2 = B + C - n * dm * dm;
    if (m2 < 0.0) m2 = 0.0;
    mean_out[(int64_t)j * dim + c] = (float)mean;
    // biased variance clamped like langevin_dynamics.py:176-178; a single chain has none (:179-181, hmc.py:300-303)
    var_out[(int64_t)j * dim + c] = n_chains > 1 ? clamp_nanprop((float)(m2 / n), 1e-10f, 1e10f) : 0.0f;
  }
  if (threadIdx.x == 0) {
    const double es = atomicExch(&wrow[3 * dim], 0.0);
    const double as = atomicExch(&wrow[3 * dim + 1], 0.0);
    if (energy_out) energy_out[j] = (float)(es / n);
    if (accept_out) accept_out[j] = (float)(as / n);
    *reinterpret_cast<unsigned long long*>(wrow + 3 * dim + 2) = 0ull;
  }
}

}  // namespace

int launch_diag_finish(const float* partials, int32_t n_kept, int64_t n_blocks, int32_t S, int32_t E, int64_t n_chains,
                       int32_t dim, float* mean_out, float* var_out, float* energy_out, float* accept_out, double* work,
                       hipStream_t st) {
  int K = 1;
  if (E < 0) {  // records of interleaved classes (diag.h plan_classes)
    E = -E;
    K = diag::diag_classes(dim);
  }
  const int W = E % dim == 0 ? 1 : dim / E;
  const int64_t n_win = ceil_div64(n_blocks, W);
  int64_t Q = ceil_div64(n_win, 64);  // >= 64 records per workgroup and slot
  const int64_t cap = ceil_div64(256 * 8, (int64_t)n_kept * W);  // ~ 8 workgroups per CU in total
  if (Q > cap) Q = cap;
  if (Q < 1) Q = 1;
  if (Q > 65535) Q = 65535;
  hipLaunchKernelGGL(diag_finish_kernel, dim3((unsigned)n_kept, (unsigned)W, (unsigned)Q), dim3(kBlock), 0, st, partials,
                     n_blocks, S, E, n_chains, dim, mean_out, var_out, energy_out, accept_out, work, K);
  return check_launch("ebm_diag_finish_f32");
}

// Per-class issue probes (bench.py prices the lean Langevin loop with them, in the run, on the box): the same
// dependent-free shape as probe_valu_kernel with the op under test in the stream.
//   kind 1: v_mad_u64_u32 + v_xor_b32 per slot (the Philox multiply)      kind 2: v_log_f32 + v_add_f32 (a transcendental)
//   kind 3: v_pk_fma_f32, two slots per instruction (5: all-VGPR operands, 6: v_pk_mul_f32)   kind 4: v_bitop3_b32 (the three-input xor)
template <int KIND>
__global__ __launch_bounds__(kBlock) void probe_issue_kernel(float* __restrict__ out, int iters) {
  if constexpr (KIND == 1 || KIND == 4) {
    uint32_t a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = threadIdx.x * 2654435761u + j;
    uint32_t k1 = 0x9E3779B9u + blockIdx.x, k2 = 0xBB67AE85u + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        if constexpr (KIND == 1) {
          const uint64_t p = (uint64_t)0xD2511F53u * a[j];
          a[j] = (uint32_t)(p >> 32) ^ (uint32_t)p;
        } else {
          a[j] = __builtin_amdgcn_bitop3_b32(a[j], k1, k2, 0x96);
          asm volatile("" : "+v"(a[j]));  // keep one instruction per slot (x ^ k1 ^ k2 twice is x)
        }
      }
    }
    uint32_t s = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[(int64_t)blockIdx.x * kBlock + threadIdx.x] = __builtin_bit_cast(float, s);
  } else if constexpr (KIND == 2) {
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.3f + threadIdx.x * 1e-4f + j * 0.01f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = __builtin_amdgcn_logf(a[j]) + 2.0f;
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j];
    out[(int64_t)blockIdx.x * kBlock + threadIdx.x] = s;
  } else {
    // packed f32: EIGHT independent chains, 32 instructions per trip of the loop (the trip's scalar bookkeeping and the
    // dependent-issue distance are then out of the measurement).  kind 3: multiplicand in an SGPR pair (how the mixture
    // kernels feed their means), kind 5: every operand a VGPR pair, kind 6: v_pk_mul_f32.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = (f32x2){threadIdx.x * 1e-3f + j, threadIdx.x * 2e-3f + j};
    f32x2 ms = {1.0001f, 0.9999f};
    asm volatile("" : "+s"(ms));
    f32x2 mv = {1.0001f + threadIdx.x * 1e-9f, 0.9999f}, cv = {0.5f, 0.25f + threadIdx.x * 1e-9f};
    asm volatile("" : "+v"(mv), "+v"(cv));
    for (int it = 0; it < iters; it += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if constexpr (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "s"(ms), "v"(cv));
          else if constexpr (KIND == 5) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(mv), "v"(cv));
          else asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[j]) : "v"(mv));
        }
    }
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += a[j].x + a[j].y;
    out[(int64_t)blockIdx.x * kBlock + threadIdx.x] = s;
  }
}

// kind 7: the lean Langevin loop's own static instruction mix per float4 group and step (langevin_elem.h; scripts/isa_mix.py:
// 18 v_mad_u64_u32, 20 v_bitop3_b32, 8 transcendentals, 20 packed-f32 (14 v_pk_mul + 6 v_pk_add), 10 plain = 76 instructions;
// SQ_INSTS_VALU of the kernel: 76.1 per group-step), dependency-free: every
// register chain is touched at most twice per trip, ~40 instructions apart.  At eight waves per SIMD its duration per trip
// is the ceiling a kernel made of exactly this mix can reach on this box at this clock -- bench.py reports the lean kernel's
// time per group-step against it (roofline.valu.frac_of_mixed_ceiling, <= 1 by construction).
__global__ __launch_bounds__(kBlock, 8) void probe_mix_kernel(float* __restrict__ out, int iters) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  uint32_t u[10];
  float f[8], g[3];
  f32x2 v[10];
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    u[j] = threadIdx.x * 2654435761u + j;
    v[j] = (f32x2){threadIdx.x * 1e-3f + j, threadIdx.x * 2e-3f + j};
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) f[j] = 0.3f + threadIdx.x * 1e-4f + j * 0.01f;
#pragma unroll
  for (int j = 0; j < 3; ++j) g[j] = threadIdx.x * 1e-3f + j;
  uint32_t k1 = 0x9E3779B9u + blockIdx.x, k2 = 0xBB67AE85u + threadIdx.x;
  f32x2 mv = {1.0001f + threadIdx.x * 1e-9f, 0.9999f}, cv = {0.5f, 0.25f + threadIdx.x * 1e-9f};
  float m = 1.0001f + threadIdx.x * 1e-9f, c = 0.5f;
  asm volatile("" : "+v"(mv), "+v"(cv), "+v"(m), "+v"(c), "+v"(k1), "+v"(k2));
  for (int it = 0; it < iters; it += 2) {  // two trips per pass of the loop, like the kernel's two steps per trip
#pragma unroll
    for (int h = 0; h < 4; ++h) {       // half trips: 9 multiplies + 10 three-input xors + 4 transcendentals + 10 packed each
#pragma unroll
      for (int j = 0; j < 10; ++j) {
        if (j < 9) {
          const uint64_t p = (uint64_t)0xD2511F53u * u[j];                                   // v_mad_u64_u32 (v_mul_hi + v_mul_lo)
          u[j] = __builtin_amdgcn_bitop3_b32((uint32_t)(p >> 32), (uint32_t)p, k1, 0x96);  // v_bitop3_b32
        } else {
          u[j] = __builtin_amdgcn_bitop3_b32(u[j], k1, k2, 0x96);
        }
        asm volatile("" : "+v"(u[j]));
        if (j < 7) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(mv));  // 14 v_pk_mul_f32 + 6 v_pk_add_f32 per trip, as in the loop
        else asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(cv));
        if (j < 4) {
          f[4 * (h & 1) + j] = __builtin_amdgcn_logf(f[4 * (h & 1) + j]) + 2.0f;             // transcendental + one plain op
          asm volatile("" : "+v"(f[4 * (h & 1) + j]));
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(g[j % 3]) : "v"(m), "v"(c));  // 8 + 2 = 10 plain per trip
  }
  uint32_t su = 0;
  float s = 0.0f;
#pragma unroll
  for (int j = 0; j < 10; ++j) {
    su += u[j];
    s += v[j].x + v[j].y;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) s += f[j];
#pragma unroll
  for (int j = 0; j < 3; ++j) s += g[j];
  out[(int64_t)blockIdx.x * kBlock + threadIdx.x] = s + __builtin_bit_cast(float, su & 0x007fffffu);
}

int launch_probe_issue(float* out, int32_t blocks, int32_t iters, int32_t kind, hipStream_t st) {
  const dim3 g((unsigned)blocks), b(kBlock);
  switch (kind) {
    case 0: hipLaunchKernelGGL(probe_valu_kernel, g, b, 0, st, out, iters); break;
    case 1: hipLaunchKernelGGL(probe_issue_kernel<1>, g, b, 0, st, out, iters); break;
    case 2: hipLaunchKernelGGL(probe_issue_kernel<2>, g, b, 0, st, out, iters); break;
    case 3: hipLaunchKernelGGL(probe_issue_kernel<3>, g, b, 0, st, out, iters); break;
    case 4: hipLaunchKernelGGL(probe_issue_kernel<4>, g, b, 0, st, out, iters); break;
    case 5: hipLaunchKernelGGL(probe_issue_kernel<5>, g, b, 0, st, out, iters); break;
    case 6: hipLaunchKernelGGL(probe_issue_kernel<6>, g, b, 0, st, out, iters); break;
    case 7: hipLaunchKernelGGL(probe_mix_kernel, g, b, 0, st, out, iters); break;
    default: return fail(EBM_EKIND, "ebm_probe_issue_f32: kind %d", kind);
  }
  return check_launch("ebm_probe_issue_f32");
}

int launch_probe_valu(float* out, int32_t blocks, int32_t iters, hipStream_t st) {
  hipLaunchKernelGGL(probe_valu_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, out, iters);
  return check_launch("ebm_probe_valu_f32");
}

int launch_noise_fill(float* out, int64_t n_elem, int32_t kind, uint64_t seed, uint64_t offset,
                      const uint64_t* rng_dev, hipStream_t st) {
  const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32)};
  hipLaunchKernelGGL(noise_fill_kernel, dim3(grid_for(ceil_div64(n_elem, 4))), dim3(kBlock), 0, st,
                     out, n_elem, kind, key, offset, rng_dev);
  return check_launch("ebm_noise_fill_f32");
}

int launch_leapfrog_kick_drift(const float* x, const float* p, const float* force, float* x_new,
                               float* p_half, int64_t n_chains, int32_t dim, float eps,
                               int32_t mass_kind, double mass_scalar, const float* mass_diag,
                               int32_t safe, hipStream_t st) {
  const int64_t n = n_chains * (int64_t)dim;
  const float half_eps = 0.5f * eps;
  const float safe_mass = (float)(mass_scalar < 1e-10 ? 1e-10 : mass_scalar);  // max(mass, 1e-10)
  hipLaunchKernelGGL(kick_drift_kernel, dim3(grid_for(n)), dim3(kBlock), 0, st, x, p, force, x_new,
                     p_half, n, dim, eps, half_eps, mass_kind, safe_mass, mass_diag, safe);
  return check_launch("ebm_leapfrog_kick_drift_f32");
}

int launch_leapfrog_kick(float* x_new, const float* p_half, const float* force, float* p_new,
                         int64_t n_elem, float eps, int32_t safe, hipStream_t st) {
  const float half_eps = 0.5f * eps;
  hipLaunchKernelGGL(kick_kernel, dim3(grid_for(n_elem)), dim3(kBlock), 0, st, x_new, p_half, force,
                     p_new, n_elem, half_eps, safe);
  return check_launch("ebm_leapfrog_kick_f32");
}

int launch_descent_step(const float* x, const float* grad, float* v, float* out, int64_t n_elem, float eta,
                        float momentum, hipStream_t st) {
  hipLaunchKernelGGL(descent_step_kernel, dim3(grid_for(n_elem)), dim3(kBlock), 0, st, x, grad, v, out, n_elem,
                     -eta, momentum);
  return check_launch("ebm_descent_step_f32");
}

int launch_lookahead(const float* x, const float* v, float* out, int64_t n_elem, float momentum, hipStream_t st) {
  hipLaunchKernelGGL(lookahead_kernel, dim3(grid_for(n_elem)), dim3(kBlock), 0, st, x, v, out, n_elem, momentum);
  return check_launch("ebm_lookahead_f32");
}

// bit v of out[0]: the component means differ somewhere in columns 4v .. 4v + 3 (the EBM_ENERGY_GMM hint; one workgroup)
__global__ __launch_bounds__(kBlock) void gmm_active_columns_kernel(const float* __restrict__ means, int32_t n_comp, int32_t dim,
                                                                    int32_t* __restrict__ out) {
  __shared__ unsigned mask;
  if (threadIdx.x == 0) mask = 0u;
  __syncthreads();
  unsigned mine = 0u;
  for (int i = threadIdx.x; i < n_comp * dim; i += kBlock) {
    const int c = i % dim;
    if (means[i] != means[c]) mine |= 1u << (c >> 2);
  }
  if (mine) atomicOr(&mask, mine);
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (int32_t)mask;
}

int launch_gmm_active_columns(const float* means, int32_t n_comp, int32_t dim, int32_t* out, hipStream_t st) {
  hipLaunchKernelGGL(gmm_active_columns_kernel, dim3(1), dim3(kBlock), 0, st, means, n_comp, dim, out);
  return check_launch("ebm_gmm_active_columns_i32");
}

int launch_pcd_gather(const float* buffer, int64_t buffer_size, int32_t dim, float* out, int64_t batch, int64_t stride,
                      const int64_t* offsets, int64_t* rows_out, uint64_t seed, uint64_t offset, const uint64_t* rng_dev,
                      hipStream_t st) {
  const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32)};
  hipLaunchKernelGGL(pcd_gather_kernel, dim3(grid_for(batch * dim)), dim3(kBlock), 0, st, buffer, buffer_size, dim, out,
                     batch, stride, offsets, rows_out, key, offset, rng_dev);
  return check_launch("ebm_pcd_gather_f32");
}

int launch_pcd_start_points(const float* buffer, int64_t buffer_size, int32_t dim, float* out, int64_t batch, int64_t stride,
                            int64_t n_noise, float noise_scale, uint64_t seed, uint64_t step, const uint64_t* rng_dev, hipStream_t st) {
  const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32)};
  int bits = 2;  // the Feistel network needs a bit on either side
  while (bits < 31 && (1LL << bits) < batch) ++bits;
  hipLaunchKernelGGL(pcd_start_points_kernel, dim3(grid_for(batch * dim)), dim3(kBlock), 0, st, buffer, buffer_size, dim, out, batch,
                     stride, n_noise, noise_scale, bits, key, step, rng_dev);
  return check_launch("ebm_pcd_start_points_f32");
}

int64_t cd_loss_work_bytes() { return (int64_t)kCdLossBlocks * 4 * sizeof(double) + 16; }
int launch_cd_loss(const float* e_both, int64_t n, float reg, void* work, float* loss_out, float* finite_out, hipStream_t st) {
  int64_t blocks = ceil_div64(n, (int64_t)kBlock * 4);
  if (blocks > kCdLossBlocks) blocks = kCdLossBlocks;
  if (blocks < 1) blocks = 1;
  double* partials = (double*)work;
  unsigned int* ticket = (unsigned int*)((char*)work + (int64_t)kCdLossBlocks * 4 * sizeof(double));
  hipLaunchKernelGGL(cd_loss_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, e_both, n, reg, partials, ticket, loss_out, finite_out);
  return check_launch("ebm_cd_loss_f32");
}
int launch_cd_loss_seed(const float* e_both, int64_t n, float reg, const float* upstream, const float* finite, float* seed_out,
                        hipStream_t st) {
  hipLaunchKernelGGL(cd_loss_seed_kernel, dim3(grid_for(2 * n)), dim3(kBlock), 0, st, e_both, n, reg, upstream, finite, seed_out);
  return check_launch("ebm_cd_loss_backward_f32");
}

int launch_pcd_scatter(float* buffer, int64_t buffer_size, int32_t dim, const float* samples, int64_t batch,
                       int64_t write_pos, const int64_t* pos_dev, hipStream_t st) {
  hipLaunchKernelGGL(pcd_scatter_kernel, dim3(grid_for(batch * dim)), dim3(kBlock), 0, st, buffer, buffer_size, dim,
                     samples, batch, write_pos, pos_dev);
  return check_launch("ebm_pcd_scatter_f32");
}

int launch_hmc_accept(float* x, const float* x_prop, const float* h0, const float* h1,
                      const float* u, uint8_t* mask, uint32_t* count, int64_t n_chains, int32_t dim,
                      uint64_t seed, uint64_t offset, const uint64_t* rng_dev, hipStream_t st) {
  const RngKey key{(uint32_t)seed, (uint32_t)(seed >> 32)};
  hipLaunchKernelGGL(hmc_accept_kernel, dim3(grid_for(n_chains)), dim3(kBlock), 0, st, x, x_prop, h0,
                     h1, u, mask, count, n_chains, dim, key, offset, rng_dev);
  return check_launch("ebm_hmc_accept_f32");
}

int launch_chain_stats(const float* x, int64_t n_chains, int32_t dim, float* mean_out,
                       float* var_out, double* work, hipStream_t st) {
  if (dim >= 4 && dim <= 1024 && (dim & (dim - 1)) == 0 && n_chains * (int64_t)dim >= 1024) {
    int64_t blocks = ceil_div64(n_chains * (int64_t)dim / 4, (int64_t)kBlock * 16);  // >= 16 float4 per lane
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(chain_stats_wide_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, st, x, n_chains, dim, work,
                       mean_out, var_out);
    return check_launch("ebm_chain_stats_f32");
  }
  const int gx = (dim + kStatCols - 1) / kStatCols;
  int64_t gy = ceil_div64(n_chains, kStatRows * 64);  // >= 64 rows per row-lane
  if (gy < 1) gy = 1;
  const int64_t cap = (256 * 8 + gx - 1) / gx;
  if (gy > cap) gy = cap;
  const dim3 grid(gx, (unsigned)gy);
  hipLaunchKernelGGL(chain_stats_kernel, grid, dim3(kBlock), 0, st, x, n_chains, dim, work, mean_out, var_out);
  return check_launch("ebm_chain_stats_f32");
}

}  // namespace ebm
