// Kernel body of the wide-MLP kernels -- see mlp_wide.hip for the design.  Three translation units instantiate it so that
// they compile in parallel: mlp_wide.hip (H = 64 / 128, weights in LDS), mlp_stream.hip (H = 256, weights streamed from
// L2), mlp_wide_hmc.hip (the HMC transition kernel).  The workgroup set-up and the evaluation itself are text shared by
// the chain kernel here and the HMC state machine: mlp_wide_setup.inc, mlp_wide_eval.inc.
#pragma once
#include "ebm_common.h"
#include "gauss_bf16x3.h"  // static_for
#include "mlp_b16.h"
#include "diag.h"

namespace ebm {
namespace widemlp {

#ifndef EBM_MLP_KBLOCK
#define EBM_MLP_KBLOCK 256  // (512: scripts/experiments -- two waves per SIMD sharing one image)
#endif
constexpr int kBlock = EBM_MLP_KBLOCK;

// How a shape keeps its weights (MODE of the kernels): 0 = fp32 in LDS, exact-f32 MFMA; 1 = STREAM (H = 256: fp32 read from
// L2, exact-f32 MFMA); 2 = B16 (three bf16 split images in LDS, bf16 MFMA at fp32 accuracy: mlp_b16.h) wherever the images fit
// the 160 KiB: H = 64 at every input width, H = 128 up to dim 64; 3 = SLAB (round 4: H = 128, dim 65 .. 128 -- the W2 image in
// LDS, the W1 image PRE-SPLIT in global memory and walked slab by slab through two 24 KB LDS buffers: mlp_b16.h "MODE 3"),
// chosen at run time when the caller hands over the image (WideArgs::w1_image); 4 = THIN (round 5: dim <= 2 -- config 5's shape --: B16 for
// W2, while the two contractions with W1 are 2 FMAs per hidden unit and run on the vector unit in exact fp32 from a transposed fp32
// copy of W1 in LDS: no operand splits of x and d1, no W1 image, 96 MFMAs and ~230 vector instructions fewer per evaluation in a
// kernel whose cost is its instruction count).  -DEBM_MLP_F32LDS (scripts only): round 2.
__host__ __device__ constexpr int b16_cols(int dt) { return dt == 1 ? 32 : (dt == 2 ? 64 : 128); }  // width of the W1 image
__host__ __device__ constexpr int wide_mode(int ht, int dt) {
#ifdef EBM_MLP_F32LDS
  return ht > 4 ? 1 : 0;
#else
  return ht > 4 ? 1 : ((ht == 2 || dt <= 2) ? 2 : 0);
#endif
}
__host__ __device__ constexpr size_t wide_smem_bytes(int ht, int dt, int mode) {
  const int H = 32 * ht, DP = 32 * dt;
  return mode == 3 ? (size_t)3 * H * sizeof(float) + 2 * mlpb16::kSlabBytes + mlpb16::image_bytes(H, H)
         : mode == 4 ? (size_t)7 * H * sizeof(float) + mlpb16::image_bytes(H, H)
         : mode == 1 ? (size_t)(16 * ht * kBlock + 3 * H) * sizeof(float)
         : mode == 2 ? (size_t)3 * H * sizeof(float) + mlpb16::image_bytes(H, H) + mlpb16::image_bytes(H, b16_cols(dt))
                     : (size_t)(H * (H + 1) + H * (DP + 1) + 3 * H) * sizeof(float);
}
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const f32x4 __attribute__((address_space(1))) * gptr4;  // 16-byte global loads of the STREAM variant

struct WideArgs {
  float* x;              // [n, dim] in/out (k_steps > 0) or input (k_steps == 0)
  int64_t n_chains;
  int32_t dim;
  int32_t k_steps;
  float eta, sqrt_eta, noise_coef;
  const float4* table;
  int clamp_on;
  float cmin, cmax;
  int32_t thin, n_kept;
  float* traj;
  const float* noise;    // [k, n, dim] or null
  RngKey key;
  uint64_t step0;
  const float* params;   // packed W1[H,dim] b1[H] W2[H,H] b2[H] w3[H] b3[1]
  float* energy_out;     // k_steps == 0: E(x)[n]
  float* grad_out;       // k_steps == 0: dE/dx[n, dim]
  float* diag_partials;  // in-kernel diagnostics records (one per WAVE: 32 chains), or null
  int64_t diag_blocks;   // records per kept step = ceil(n_chains / 32)
  const char* w1_image;  // MODE 3: the pre-split W1 image (ebm_mlp_w1_image_f32), or null
  const uint64_t* rng_dev;  // ebm_langevin_chain_dev_f32: {seed, step} in device memory (step0 is then an offset from it), or null
  const float* seed;        // FAST = 3 (ebm_mlp_backward_acts_f32): dL/dE[n] of a training backward, or null (= 1)
  float* acts;              // FAST = 3: the activation planes, tiled: [act_stride / 32][3][H][32] (h1 | a2 | d1 of 32 rows, hidden-major)
  int64_t act_stride;       // ... the padded row count: n_chains rounded up to whole workgroups of 128 chains (no lane, no wave needs masking)
};

extern __shared__ __attribute__((aligned(16))) float wide_smem[];

// In-kernel diagnostics records: diag::wave_record / diag::wave_record_tail (diag.h) -- one record per WAVE of 32 chains.
template <int DT>
__device__ __noinline__ void wave_record(float* partials, int64_t n_blocks, int keep, int64_t wave_id, int dim, const float (&xs)[DT][16],
                                         bool active, int lane) {
  diag::wave_record<DT>(partials, n_blocks, keep, wave_id, dim, [&](int td, int r) { return xs[td][r]; }, active, lane);
}
using diag::wave_record_tail;

// -DEBM_PHASE_TIMES (scripts/mlp_phase_times.py only): wave 0 of workgroup 0 logs the shader clock at the phase boundaries of
// every evaluation (stamps cost a drained LDS queue each: read the phases relative to each other, not against a plain run).
#ifdef EBM_PHASE_TIMES
__device__ unsigned long long ebm_phase_log[8192];
#define EBM_STAMP()                                                                                     \
  do {                                                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    if (__builtin_amdgcn_readfirstlane(blockIdx.x) == 0 && __builtin_amdgcn_readfirstlane(threadIdx.x) == 0) { \
      const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                      \
      if (stamp_n_ < 8192 && (threadIdx.x & 63) == 0) ebm_phase_log[stamp_n_] = now_;                     \
      ++stamp_n_;                                                                                        \
    }                                                                                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
  } while (0)
#else
#define EBM_STAMP() do {} while (0)
#endif

__device__ __forceinline__ float sigmoid_fast(float a) { return __builtin_amdgcn_rcpf(1.0f + __expf(-a)); }
__device__ __forceinline__ constexpr int row_of(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// out[t] += A_t * B over NK x 16 K-steps; the LDS operands of step s + 1 are requested before the MFMAs of step s
// issue, so their latency hides under NT x 64 matrix-pipe cycles.  addr(t, tk, r): LDS word of A for output tile t at
// K-step (tk, r) (this lane's row / K-half folded in by the caller); bval(tk, r): this lane's B value.
template <int NT, int NK, class Addr, class Bval>
__device__ __forceinline__ void contract(f32x16 (&out)[NT], const float* lds, Addr addr, Bval bval) {
  float cur[NT], nxt[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) cur[t] = lds[addr(t, 0, 0)];
#pragma unroll
  for (int tk = 0; tk < NK; ++tk)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int s = tk * 16 + r;
      if (s + 1 < NK * 16) {
#pragma unroll
        for (int t = 0; t < NT; ++t) nxt[t] = lds[addr(t, (s + 1) >> 4, (s + 1) & 15)];
      }
      const float b = bval(tk, r);
#pragma unroll
      for (int t = 0; t < NT; ++t) out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur[t], b, out[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);  // keep the issue order: next loads, this step's MFMAs
#pragma unroll
      for (int t = 0; t < NT; ++t) cur[t] = nxt[t];
    }
}

using gauss3::static_for;

// STREAM, forward walk: ld4(t, tk, q) -> the four A words of output tile t at K-steps (tk, 4 q .. 4 q + 3) (one 16-byte
// global load: lane (m, h) takes bytes 32 q + 16 h .. + 15 of the 128-byte segment of weight row 32 t + m).  A stage
// is one (t, tk): its four loads go out back to back, so every 128-byte line they touch is fetched from L2 once (a
// stage built across tiles instead re-fetched each line four times -- the L1 does not hold 4 waves x 16 KB), and feed
// 16 accumulations into ONE tile (the matrix pipe forwards a back-to-back accumulator); requested PF stages ahead.
struct NoFix {
  template <class... A>
  __device__ __forceinline__ float operator()(float v, A...) const { return v; }
};

template <int NT, int NK, int PF, class Ld4, class Bval, class Fix = NoFix>
__device__ __forceinline__ void contract_rows(f32x16 (&out)[NT], Ld4 ld4, Bval bval, Fix fix = Fix{}) {
  constexpr int NS = NT * NK;
  f32x4 buf[PF + 1][4];
  static_for<(PF < NS ? PF : NS)>([&](auto s_) __attribute__((always_inline)) {
    constexpr int s = decltype(s_)::value;
    static_for<4>([&](auto q_) __attribute__((always_inline)) {
      buf[s % (PF + 1)][decltype(q_)::value] = ld4(std::integral_constant<int, s / NK>{}, std::integral_constant<int, s % NK>{}, q_);
    });
  });
  static_for<NS>([&](auto s_) __attribute__((always_inline)) {
    constexpr int s = decltype(s_)::value;
    if constexpr (s + PF < NS) {
      constexpr int n = s + PF;
      static_for<4>([&](auto q_) __attribute__((always_inline)) {
        buf[n % (PF + 1)][decltype(q_)::value] = ld4(std::integral_constant<int, n / NK>{}, std::integral_constant<int, n % NK>{}, q_);
      });
    }
    constexpr int t = s / NK, tk = s % NK;
    static_for<16>([&](auto r_) __attribute__((always_inline)) {
      constexpr int r = decltype(r_)::value;
      // fix(): what has to happen to a loaded word (zeroing a padding column) happens HERE, at its use -- applied where the
      // load is issued it would make every stage wait for the request it has just made
      out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fix(buf[s % (PF + 1)][r >> 2][r & 3], std::integral_constant<int, tk>{}, r_), bval(tk, r),
                                                    out[t], 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// STREAM, transposed walk: ld1(t, tk, r) -> the A word of output tile t at K-step (tk, r) (one coalesced dword load);
// a stage = one K-step = NT loads feeding NT MFMAs, requested PF stages ahead.
template <int NT, int NK, int PF, class Ld1, class Bval, class Fix = NoFix>
__device__ __forceinline__ void contract_cols(f32x16 (&out)[NT], Ld1 ld1, Bval bval, Fix fix = Fix{}) {
  constexpr int NS = NK * 16;
  float buf[PF + 1][NT];
  static_for<(PF < NS ? PF : NS)>([&](auto s_) __attribute__((always_inline)) {
    constexpr int s = decltype(s_)::value;
    static_for<NT>([&](auto t_) __attribute__((always_inline)) {
      constexpr int t = decltype(t_)::value;
      buf[s % (PF + 1)][t] = ld1(std::integral_constant<int, t>{}, std::integral_constant<int, s / 16>{}, std::integral_constant<int, s & 15>{});
    });
  });
  static_for<NS>([&](auto s_) __attribute__((always_inline)) {
    constexpr int s = decltype(s_)::value;
    if constexpr (s + PF < NS) {
      constexpr int n = s + PF;
      static_for<NT>([&](auto t_) __attribute__((always_inline)) {
        constexpr int t = decltype(t_)::value;
        buf[n % (PF + 1)][t] = ld1(std::integral_constant<int, t>{}, std::integral_constant<int, n / 16>{}, std::integral_constant<int, n & 15>{});
      });
    }
    const float b = bval(s / 16, s & 15);
    static_for<NT>([&](auto t_) __attribute__((always_inline)) {
      constexpr int t = decltype(t_)::value;
      out[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fix(buf[s % (PF + 1)][t], t_), b, out[t], 0, 0, 0);
    });
    __builtin_amdgcn_sched_barrier(0);
  });
}

// FAST = 1: the plain call only -- a real chain (k_steps > 0), in-kernel noise, no clamp, dim % 4 == 0 or dim == 2, no diagnostics records --
// with everything else compiled out (the injected-noise and per-element Philox paths, the clamp, the record code and its
// out-of-line call): the update loses its uniform branches, the kernel a third of its code.  FAST = 2: the same call WITH records
// (return_diagnostics=True otherwise falls to the general kernel, 8 % slower: 512 registers and spills where this one has neither).
// Workgroups per CU: H = 128 -- one (the W2 image alone is 96 KB, an evaluation holds 350 - 512 registers).  H = 64 at dim <= 64: TWO (the
// images are 36 - 48 KB and the FAST evaluation fits 256 registers without a spill, so the second workgroup's waves fill the issue
// slots a lone wave leaves empty: one instruction per ~6.5 cycles alone, ~3.1 with two waves on a SIMD).
__host__ __device__ constexpr int wide_min_blocks(int ht, int dt, int mode, int fast) {
  return (ht == 2 && dt <= 2 && (mode == 2 || mode == 4) && fast == 1) ? 2 : 1;
}
template <int HT, int DT, int MODE, int FAST = 0>
__global__ __launch_bounds__(kBlock, wide_min_blocks(HT, DT, MODE, FAST)) void mlp_wide_chain_kernel(WideArgs a) {
  // (round 6) THIN chain calls: the layers' pre-activations are held PRE-SCALED by -log2 e (b1, b2 and the forward copy of W1 are
  // staged times -log2 e; layer 2 needs nothing -- its input silu(a1) comes out times -log2 e as well and W2 (L h1) + L b2 = L a2),
  // so that sigmoid(a) = 1 / (1 + exp2(a')) costs no multiply per element: 128 vector instructions per evaluation fewer
  // (mlp_wide_eval_b16.inc `EVAL_SCALED`).  The training forward (FAST = 3) stores activation planes and keeps the plain form.
  constexpr bool EVAL_SCALED = MODE == 4 && (FAST == 1 || FAST == 2);
#include "mlp_wide_setup.inc"

  // Evaluation-only launches (k_steps == 0: energies / gradients, the training backward) come back here for their next tile of
  // 32 chains per wave: the grid is one workgroup per CU and each sweeps the batch, so the weight images are staged 256 times per
  // launch instead of once per 128 rows (staging was 40 of the 65 us a forward pass over 131 072 rows took).
tile_again:
  // the state in the C/D layout: xr[td][r] = x[sample][32 td + row_of(r, h)], zero beyond dim
  float xr[DT][16];
#pragma unroll
  for (int td = 0; td < DT; ++td)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int c0 = 32 * td + 8 * q + 4 * h;
      if (MODE != 4 && quads && active && c0 + 3 < dim) {
        const float4 v = *reinterpret_cast<const float4*>(a.x + sample * dim + c0);
        xr[td][4 * q] = v.x; xr[td][4 * q + 1] = v.y; xr[td][4 * q + 2] = v.z; xr[td][4 * q + 3] = v.w;
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)  // (THIN: columns 0 and 1 are all there is -- the other registers are the constant 0)
          xr[td][4 * q + i] = (MODE == 4 && (q > 0 || i > 1)) ? 0.0f : ((active && c0 + i < dim) ? a.x[sample * dim + c0 + i] : 0.0f);
      }
    }

  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  // RNG coordinates: by value, or (a launch captured in a HIP graph) read from device memory -- wave-uniform, two scalar loads
  RngKey rkey = a.key;
  uint64_t rstep0 = a.step0;
  if (a.rng_dev) {
    const uint64_t seed_dev = a.rng_dev[0];
    rkey = RngKey{(uint32_t)seed_dev, (uint32_t)(seed_dev >> 32)};
    rstep0 += a.rng_dev[1];
  }
  // Diagnostics: the reference reports the mean energy of the KEPT state, i.e. of x after the update; that energy is what
  // the NEXT step's evaluation computes anyway, so the record's energy share is written one evaluation late and only a
  // kept LAST step costs an evaluation of its own (the loop runs one more time, without an update).
  const int64_t wave_id = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
  [[maybe_unused]] int diag_keep = 0;
  int diag_pending = -1;
  const bool diag_tail = FAST != 1 && a.diag_partials && a.k_steps > 0 && a.thin > 0 && a.k_steps % a.thin == 0;
  const int n_evals = FAST == 1 ? a.k_steps : (a.k_steps > 0 ? a.k_steps + (diag_tail ? 1 : 0) : 1);

#ifdef EBM_PHASE_TIMES
  int stamp_n_ = 0;
#endif
  for (int step = 0; step < n_evals; ++step) {
    EBM_STAMP();
    constexpr bool eval_block_cuts = FAST != 0;
    constexpr bool eval_pin = true;
    constexpr bool eval_need_energy = FAST != 1;  // (FAST = 1: no records, no evaluation-only launch -- nobody reads the energy)
    constexpr bool eval_store_acts = FAST == 3;  // the backward pass of a training step: seed-scaled, activations stored (eval_b16.inc)
    // the wave's tile of the activation planes: [n_pad / 32][3][H][32] floats -- one contiguous 12 H-float block per tile of 32 rows
    [[maybe_unused]] float* act_base = FAST == 3 ? a.acts + (size_t)__builtin_amdgcn_readfirstlane((int)((sample - m) >> 5)) * (size_t)(3 * H * 32) : nullptr;
    if constexpr (FAST == 3) asm volatile("" : "+s"(act_base));  // (the store addresses are formed at their stores, not hoisted)
    [[maybe_unused]] const uint32_t act_lane = (uint32_t)((m + 128 * h) * sizeof(float));  // its column of the tile + its half's four rows
    [[maybe_unused]] const float act_seed = (FAST == 3 && active && a.seed) ? a.seed[sample] : 1.0f;  // scales grad_out (the planes are seed-free)
    bool eval_energy_only = FAST != 1 && a.k_steps > 0 && step >= a.k_steps;  // the extra evaluation of a kept last step
    // (evaluation only, and nothing but the energy asked for: the forward pass alone -- the energies of a training forward)
    if (FAST == 0 && a.k_steps == 0 && !a.grad_out) eval_energy_only = true;
    // FAST: never -- but left as an opaque (always false) scalar: the branch it guards cuts the evaluation's one basic block in
    // two, and without that cut the scheduler stretches live ranges until 160 registers spill (19 with it)
    if constexpr (FAST == 1) {
      int never = 0;
      asm volatile("" : "+s"(never));
      eval_energy_only = never != 0;
    }
    [[maybe_unused]] const bool slab_more = step + 1 < n_evals;
    // (round 6) THIN + FAST (config 5's call): this step's noise does not depend on the evaluation, so it is drawn in 15 slots of <= 5
    // instructions behind the MFMAs that have no epilogue to carry (eval_b16.inc `eval_aux`: tile 0's twelve of each tile-major tail) --
    // Philox's ten rounds, the half of the counter this chain owns, one Box-Muller pair.  The same draws, bit for bit, as
    // normal4_at() on (chain >> 1, step) followed by the odd / even select of the general path below.
    constexpr bool aux_rng = MODE == 4 && (FAST == 1 || FAST == 2);
    [[maybe_unused]] uint32_t ax0, ax1, ax2, ax3, axk0 = rkey.k0, axk1 = rkey.k1;
    [[maybe_unused]] float axu = 0.0f, axrev = 0.0f, axr = 0.0f, eps_pre0 = 0.0f, eps_pre1 = 0.0f;
    [[maybe_unused]] const auto eval_aux = [&](auto kc) __attribute__((always_inline)) {
      constexpr int k = decltype(kc)::value;
      if constexpr (aux_rng) {
        if constexpr (k == 0) {
          int64_t sm = sample;
          asm volatile("" : "+v"(sm));  // (nothing derived from the chain index is hoisted out of the step loop)
          const uint64_t group = (uint64_t)sm >> 1, stp = rstep0 + (uint64_t)step;
          ax0 = (uint32_t)group; ax1 = (uint32_t)(group >> 32); ax2 = (uint32_t)stp; ax3 = (uint32_t)(stp >> 32);
        } else if constexpr (k <= 10) {  // one Philox round (ebm_common.h philox4x32_10): two 32 x 32 -> 64 products, two 3-input xors
          const uint64_t p0 = (uint64_t)0xD2511F53u * ax0, p1 = (uint64_t)0xCD9E8D57u * ax2;
          const uint32_t n0 = xor3((uint32_t)(p1 >> 32), ax1, axk0), n2 = xor3((uint32_t)(p0 >> 32), ax3, axk1);
          ax1 = (uint32_t)p1; ax3 = (uint32_t)p0; ax0 = n0; ax2 = n2;
          axk0 += 0x9E3779B9u; axk1 += 0xBB67AE85u;
        } else if constexpr (k == 11) {  // this chain's half of the counter: (x, y) for even chains, (z, w) for odd ones
          const bool odd = (sample & 1) != 0;
          const uint32_t ua = odd ? ax2 : ax0, ub = odd ? ax3 : ax1;  // (U4{c0, c1, c2, c3} = x, y, z, w)
          axu = u01_open_low(ua);
          axrev = (float)ub * 0x1p-32f;
          asm volatile("" : "+v"(axu), "+v"(axrev));  // (slots 0 .. 11 sit in layer 2's tail, their users behind a block cut: mlp_b16.h EBM_PIN)
        } else if constexpr (k == 12) {
          axr = -1.38629436111989061883f * __builtin_amdgcn_logf(axu);
        } else if constexpr (k == 13) {
          axr = __builtin_amdgcn_sqrtf(axr);
          eps_pre0 = __builtin_amdgcn_sinf(axrev);
        } else if constexpr (k == 14) {
          eps_pre1 = axr * __builtin_amdgcn_cosf(axrev);
          eps_pre0 = axr * eps_pre0;
          asm volatile("" : "+v"(eps_pre0), "+v"(eps_pre1));
        }
      }
    };
#include "mlp_wide_eval.inc"
    if (FAST != 1 && diag_pending >= 0) {
      wave_record_tail(a.diag_partials, a.diag_blocks, diag_pending, wave_id, dim, energy, active, false, lane);
      diag_pending = -1;
    }
    if (FAST != 1 && a.k_steps > 0 && step >= a.k_steps) break;  // the extra evaluation of a kept last step

    if ((FAST == 0 || FAST == 3) && a.k_steps == 0) {  // evaluation only
      if (active) {
        if (a.energy_out && h == 0) a.energy_out[sample] = energy;
        if (a.grad_out) {
#pragma unroll
          for (int td = 0; td < DT; ++td)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int c = 32 * td + row_of(r, h);
              if (c < dim) a.grad_out[sample * dim + c] = FAST == 3 ? g[td][r] * act_seed : g[td][r];
            }
        }
      }
      if constexpr (SLAB) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (no slab transfer outlives its workgroup)
      } else {
        // the wave's next tile (no barrier in the evaluation of these modes: every wave sweeps on its own); FAST = 3 also visits
        // the padding columns of the activation arrays
        sample += (int64_t)gridDim.x * (kBlock / 64) * 32;
        const int64_t extent = FAST == 3 ? a.act_stride : a.n_chains;
        if (__builtin_amdgcn_readfirstlane((int)((sample - m) < extent))) {
          active = sample < a.n_chains;
          goto tile_again;
        }
      }
      return;
    }

    // ------------------------------------------------------------ Euler-Maruyama update (reference op order)
    int64_t smp = sample;  // STREAM / FAST: nothing derived from the chain index (counters, addresses) is hoisted out of the loop and spilled
    if constexpr (STREAM || FAST != 0) asm volatile("" : "+v"(smp));
    if (a.table) {
      const float4 tb = a.table[step];
      eta = tb.x; sqrt_eta = tb.y; noise_coef = tb.z;
    }
#pragma unroll
    for (int td = 0; td < DT; ++td)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (MODE == 4 && q > 0) continue;  // (THIN: the state is columns 0 and 1)
        const int c0 = 32 * td + 8 * q + 4 * h;
        float eps[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        // (FAST: dim % 4 == 0 or dim == 2, and dim > 32 (DT - 1) -- only the last tile has quads past dim)
        if ((FAST != 0 && td + 1 < DT) || c0 < dim) {
          if (FAST == 0 && a.noise) {
            if (active)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (c0 + i < dim) eps[i] = a.noise[((int64_t)step * a.n_chains + smp) * dim + c0 + i];
          } else if (quads) {  // the quad is exactly one Philox counter
            const F4 nrm = normal4_at(rkey, ((uint64_t)smp * (uint64_t)dim + (uint64_t)c0) >> 2, rstep0 + (uint64_t)step);
#pragma unroll
            for (int i = 0; i < 4; ++i) eps[i] = nrm.v[i];
          } else if constexpr (aux_rng) {  // dim == 2 on the THIN kernel: drawn behind the evaluation's MFMAs (eval_aux above)
            if (td == 0 && q == 0) {
              eps[0] = eps_pre0;
              eps[1] = eps_pre1;
            }
          } else if constexpr (FAST != 0) {  // dim == 2 (config 5's shape): a chain's two elements are half a Philox counter
            if (td == 0 && q == 0) {
              const F4 nrm = normal4_at(rkey, (uint64_t)smp >> 1, rstep0 + (uint64_t)step);
              const bool odd = (smp & 1) != 0;
              eps[0] = odd ? nrm.v[2] : nrm.v[0];
              eps[1] = odd ? nrm.v[3] : nrm.v[1];
            }
          } else {
            uint64_t have = ~0ull;
            F4 nrm;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint64_t e = (uint64_t)smp * (uint64_t)dim + (uint64_t)(c0 + i);
              if ((e >> 2) != have) {
                have = e >> 2;
                nrm = normal4_at(rkey, have, rstep0 + (uint64_t)step);
              }
              const int w = (int)(e & 3);
              eps[i] = w == 0 ? nrm.v[0] : (w == 1 ? nrm.v[1] : (w == 2 ? nrm.v[2] : nrm.v[3]));
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (MODE == 4 && i > 1) continue;
          const int r = 4 * q + i;
          const float x1 = xr[td][r] - eta * g[td][r];
          const float dw = eps[i] * sqrt_eta;
          float nv = x1 + noise_coef * dw;
          if (FAST == 0 && a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
          // FAST: no select -- a padding column has x = 0, g = 0 exactly (the image columns past dim are zero) and no draw,
          // so the update leaves it 0 by itself; 16 DT selects and as many loop-invariant masks (spilled to lanes) fewer.
          // (A chain whose d1 is not finite gets NaN there one step before its real columns would hand it on anyway.)
          xr[td][r] = (FAST != 0 || c0 + i < dim) ? nv : 0.0f;
        }
      }
    if ((a.traj || (FAST != 1 && a.diag_partials)) && --until_keep == 0) {
      until_keep = a.thin;
      if (a.traj && active) {
        float* dst = a.traj + smp * (int64_t)a.n_kept * dim + keep_off;
#pragma unroll
        for (int td = 0; td < DT; ++td)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int c = 32 * td + row_of(r, h);
            if (c < dim) dst[c] = xr[td][r];
          }
      }
      keep_off += dim;
#ifndef EBM_MLP_NO_DIAG  // A/B builds only: what the records cost the general kernel
      if (FAST != 1 && a.diag_partials) {
        wave_record<DT>(a.diag_partials, a.diag_blocks, diag_keep, wave_id, dim, xr, active, lane);
        diag_pending = diag_keep++;
      }
#endif
    }
  }
  if (active) {
#pragma unroll
    for (int td = 0; td < DT; ++td)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = 32 * td + row_of(r, h);
        if (c < dim) a.x[sample * dim + c] = xr[td][r];
      }
  }
  if constexpr (SLAB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the slab a kept last step's energy pass left in flight
}

template <int HT, int DT, int MODE, int FAST>
int launch_variant(const WideArgs& a, hipStream_t st, const char* who) {
  constexpr bool STREAM = MODE == 1;
  const size_t smem = wide_smem_bytes(HT, DT, MODE);
  if (STREAM && (reinterpret_cast<uintptr_t>(a.params) & 15) != 0)
    return fail(EBM_EINVAL, "%s: the MLP parameter block must be 16-byte aligned", who);
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  if (attr_once.first()) {  // > 64 KiB of dynamic LDS needs the opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_wide_chain_kernel<HT, DT, MODE, FAST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  int64_t blocks = ceil_div64(a.n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "%s: too many chains for one launch", who);
  if (a.k_steps == 0 && MODE != 3) {  // evaluation only: one workgroup per CU sweeps the batch (the kernel's tile loop)
    int dev = 0, cus = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0 && blocks > cus) blocks = cus;
  }
  hipLaunchKernelGGL((mlp_wide_chain_kernel<HT, DT, MODE, FAST>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch(who);
}

// the FAST instantiations live in mlp_wide_fast.hip / mlp_wide_fast_diag.hip (their own translation units: compiled in parallel)
template <int HT, int DT>
int launch_fast(const WideArgs& a, hipStream_t st, const char* who);
template <int HT, int DT>
int launch_fast_diag(const WideArgs& a, hipStream_t st, const char* who);
#define EBM_FAST_DECL(HTV, DTV)                                                                          \
  template <> int launch_fast<HTV, DTV>(const WideArgs& a, hipStream_t st, const char* who);             \
  template <> int launch_fast_diag<HTV, DTV>(const WideArgs& a, hipStream_t st, const char* who);
EBM_FAST_DECL(2, 1) EBM_FAST_DECL(2, 2) EBM_FAST_DECL(2, 3) EBM_FAST_DECL(2, 4) EBM_FAST_DECL(4, 1) EBM_FAST_DECL(4, 2)
#undef EBM_FAST_DECL
// FAST = 3 (mlp_wide_train.hip): the evaluation as the backward pass of a training step (seed-scaled, activations stored)
template <int HT, int DT>
int launch_train(const WideArgs& a, hipStream_t st, const char* who);
template <> int launch_train<2, 1>(const WideArgs& a, hipStream_t st, const char* who);
template <> int launch_train<2, 2>(const WideArgs& a, hipStream_t st, const char* who);
template <> int launch_train<4, 1>(const WideArgs& a, hipStream_t st, const char* who);
template <> int launch_train<4, 2>(const WideArgs& a, hipStream_t st, const char* who);
// MODE 4 (mlp_wide_thin.hip): dim <= 2; fast = 1 the plain call, 2 the same with records, 3 the training forward / backward
template <int HT>
int launch_thin(const WideArgs& a, int fast, hipStream_t st, const char* who);
template <> int launch_thin<2>(const WideArgs& a, int fast, hipStream_t st, const char* who);
template <> int launch_thin<4>(const WideArgs& a, int fast, hipStream_t st, const char* who);
// MODE 3 (mlp_wide_slab.hip): H = 128, dim 65 .. 128 with the W1 image at hand; fast = 0 general, 1 plain call, 2 records
template <int DT>
int launch_slab(const WideArgs& a, int fast, hipStream_t st, const char* who);
template <> int launch_slab<3>(const WideArgs& a, int fast, hipStream_t st, const char* who);
template <> int launch_slab<4>(const WideArgs& a, int fast, hipStream_t st, const char* who);
// H = 128, dim <= 32, the plain call at two waves per SIMD (scripts/experiments/mlp_quad.hip: four waves share a chain tile) --
// a round-5 experiment that lost to the one-wave kernel (docs/design/mlp_wide.md, "Round 5"); linked by scripts/build_quad_ab.sh only
int launch_quad(const WideArgs& a, hipStream_t st, const char* who);
inline bool wide_fast_shape(const WideArgs& a) {  // (with or without records)
  return a.k_steps > 0 && !a.noise && !a.clamp_on && ((a.dim & 3) == 0 || a.dim == 2);
}
// (dim == 1 takes the general kernel: FAST draws a chain's noise as half a Philox counter, which is dim == 2's addressing)

template <int HT, int DT>
int launch_one(const WideArgs& a, hipStream_t st, const char* who) {
  constexpr int MODE = wide_mode(HT, DT);
#ifndef EBM_MLP_F32LDS
  if constexpr (HT == 4 && DT >= 3) {
    if (a.w1_image && (reinterpret_cast<uintptr_t>(a.w1_image) & 15) == 0)
      return launch_slab<DT>(a, wide_fast_shape(a) ? (a.diag_partials ? 2 : 1) : 0, st, who);
  }
#endif
#ifdef EBM_MLP_QUAD_EXPERIMENT
  if constexpr (HT == 4 && DT == 1) {
    if (wide_fast_shape(a) && !a.diag_partials && !ab_switch("EBM_MLP_NO_QUAD")) return launch_quad(a, st, who);
  }
#endif
#ifndef EBM_MLP_NO_THIN
  if constexpr (MODE == 2 && DT == 1 && (HT == 2 || HT == 4)) {
    if (a.dim <= 2 && wide_fast_shape(a) && !ab_switch("EBM_MLP_NO_THIN")) return launch_thin<HT>(a, a.diag_partials ? 2 : 1, st, who);
  }
#endif
  if constexpr (MODE == 2) {
#ifdef EBM_NO_FAST_DIAG  // A/B builds: records on the general kernel
    if (wide_fast_shape(a) && !a.diag_partials) return launch_fast<HT, DT>(a, st, who);
#elif defined(EBM_PLAIN_ON_FAST2)  // A/B builds: the plain call on the records instantiation too (same box: 3 - 9 % slower)
    if (wide_fast_shape(a)) return launch_fast_diag<HT, DT>(a, st, who);
#else
    if (wide_fast_shape(a)) return a.diag_partials ? launch_fast_diag<HT, DT>(a, st, who) : launch_fast<HT, DT>(a, st, who);
#endif
  }
  return launch_variant<HT, DT, MODE, 0>(a, st, who);
}

}  // namespace widemlp
}  // namespace ebm
