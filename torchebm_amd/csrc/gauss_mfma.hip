// k-fused Langevin chain for the dense Gaussian energy on the matrix cores (dim a multiple of 32,
// dim <= 128):  g = Ps (x - mu)  is the only dense contraction on the whole path (SURVEY.md §7 "hard
// parts", §8 a4), so it goes to the exact-f32 MFMA instead of an LDS mat-vec.
// Reference: torchebm/core/base_model.py:181-210 (energy), samplers/langevin_dynamics.py:154-185.
//
// Mapping (same idea as mlp.hip): a wavefront owns 32 chains; lane l = (m, h), m = l & 31 the chain,
// h = l >> 5 the K-half of v_mfma_f32_32x32x2_f32.  The chain state itself lives in the C/D layout of
// the 32x32 tiles: register r of tile t holds coordinate k = 32 t + (r&3) + 8 (r>>2) + 4 h of chain m,
// so that  g^T[i, m] = sum_k Ps[i, k] d[k, m]  takes its B-operand straight from the state registers
// (the K index is enumerated in the order the layout already holds it) and produces g in the very
// layout x is stored in: update, clamp, trajectory stores are register-to-register, and four
// consecutive registers are four consecutive coordinates = one Philox counter = one float4 access.
// Ps = (P + P^T)/2 is symmetric, so the A-operand Ps[i = 32 it + m][k] is read as Ps[k][32 it + m]:
// consecutive lanes, consecutive LDS banks.
#include "ebm_common.h"
#include "diag.h"
#include "gauss_bf16x3.h"
#include "gmm_bf16x3.h"

namespace ebm {
namespace {

constexpr int kBlock = 256;
typedef float f32x16 __attribute__((ext_vector_type(16)));

struct GaussArgs {
  float* x;
  int64_t n_chains;
  int32_t dim;
  int32_t k_steps;
  float eta, sqrt_eta, noise_coef;
  const float4* table;
  int clamp_on;
  float cmin, cmax;
  int32_t thin, n_kept;
  float* traj;
  const float* noise;
  RngKey key;
  uint64_t step0;
  const float* mean;  // [sub_dim]
  const float* prec;  // [sub_dim, sub_dim], symmetric
  int32_t sub_dim = 0;  // PACKED rows: dim = pack * sub_dim -- `pack` consecutive chains of a sub_dim-dimensional Gaussian are ONE
  int32_t pack = 1;     // row of the block-diagonal Gaussian kron(I_pack, Ps) (n_chains counts packed rows); else sub_dim = dim
  gmm3::Params gm;    // the mixture kernels (GKR > 0 below)
  diag::DiagArgs diag;     // per-workgroup diagnostics records at the kept steps (DIAG kernels)
  int diag_offset_floats;  // start of the diagnostics tile in dynamic LDS
};

extern __shared__ __attribute__((aligned(16))) float gauss_smem[];

// B3: the contraction on the bf16 matrix pipe with three-way split operands (gauss_bf16x3.h) -- 6/16 of the exact-f32
// MFMA's matrix time and, unlike it, concurrent with the step's Philox / Box-Muller VALU work.  B3 = false keeps the
// exact-f32 MFMA (EBM_GAUSS_F32MFMA=1: the A/B switch).
// FAST (B3 only): no injected noise, no clamp -- the step is ONE basic block: the normals of all quads are drawn
// first, then the contraction, and the scheduler is told to place ~VPM VALU instructions behind every MFMA, so that the
// wave's own Philox / Box-Muller work runs while the matrix pipe is busy (a bf16 32x32x16 MFMA occupies it for 32
// cycles; left alone the compiler issues the MFMAs back to back and the VALU work after them).
// BLOCK: threads per workgroup.  512 for the wide FAST kernels: eight waves share ONE LDS copy of the split matrix, i.e.
// two waves per SIMD where a 256-thread workgroup (one per CU: the matrix is 55 / 98 KB) leaves each SIMD one wave, which
// can issue at only 39 % of the VALU rate.  HIDE: tiles whose normals are drawn behind the MFMAs (the others are drawn
// after the contraction: their 16 registers per tile are then not live across it, which is what fits 256 VGPRs).
// GKR > 0: the energy is an isotropic Gaussian MIXTURE (gmm_bf16x3.h; GKR = its logit-register class 4 / 8 / 16 for up
// to 8 / 16 / 32 components) instead of the dense Gaussian: same state layout, same update, the gradient from
// gmm3::Mixture.
// DIAG: the in-kernel diagnostics records (diag.h) at the kept steps -- the workgroup's 128 chains go to an LDS tile
// in flat order, the energy of a kept state is one more evaluation (Gaussian: contraction + dot; mixture: the
// difference-form logsumexp).
// KT: trailing K-blocks of 16 coordinates that are padding only (dim <= 32 NT - 16 KT) and left out of the contraction
// (dims 36..48, 68..80, 100..112, 132..144: a quarter .. a tenth of the MFMAs and of the operand split).
template <int NT, bool B3, bool FAST = false, int BLOCK = 256, int HIDE = NT, int GKR = 0, bool DIAG = false, int KT = 0>
__device__ __forceinline__ void gauss_langevin_mfma_body(const GaussArgs& a) {
  constexpr int DIM = 32 * NT, KBU = 2 * NT - KT;
  static_assert(KT == 0 || (B3 && GKR == 0), "trimmed K-blocks: the dense Gaussian on the bf16 pipe");
  using Mix = gmm3::Mixture<NT, GKR == 0 ? 4 : GKR>;
  // LDS: the precision matrix -- fp32 [DIM][DIM], or its three operand-ready bf16 splits (1.5x the bytes) -- then mu
  float* Ps = gauss_smem;
  __bf16* aop = reinterpret_cast<__bf16*>(gauss_smem);
  float* mus = gauss_smem + (B3 ? (int)(gauss3::aop_bytes(NT) / sizeof(float)) : DIM * DIM);  // [DIM]
  // dim <= DIM, dim % 4 == 0: the tiles are zero-padded -- padded coordinates stay exactly 0 (d = 0, g = 0,
  // no noise) and whole register quads beyond dim are never loaded, drawn or stored
  const int dim = a.dim;
  if constexpr (GKR > 0) {
    Mix::stage(a.gm, gauss_smem, BLOCK);
  } else {
    // the precision of the (possibly packed) row: block-diagonal copies of the sub_dim x sub_dim matrix
    const int sd = a.sub_dim;
    const auto ps_at = [&](int r, int c) {
      if (r >= dim || c >= dim) return 0.0f;
      const int br = r / sd, bc = c / sd;
      return br == bc ? a.prec[(r - br * sd) * sd + (c - bc * sd)] : 0.0f;
    };
    if constexpr (B3) {
      gauss3::stage_split_matrix<NT, 2 * NT>(ps_at, aop, BLOCK);
    } else {
      for (int i = threadIdx.x; i < DIM * DIM; i += BLOCK) {
        const int r = i / DIM, c = i - r * DIM;
        Ps[i] = ps_at(r, c);
      }
    }
  }
  if constexpr (GKR == 0)
    for (int i = threadIdx.x; i < DIM; i += BLOCK) mus[i] = i < dim ? a.mean[i % a.sub_dim] : 0.0f;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int64_t chain = ((int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 32 + m;
  const bool active = chain < a.n_chains;
  const int64_t row = active ? chain * (int64_t)dim : 0;

  // state in the C/D layout; quad q of tile t = coordinates 32t + 8q + 4h .. +3
  f32x16 x[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k0 = 32 * t + 8 * q + 4 * h;
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
      if (active && k0 < dim) xv = *reinterpret_cast<const float4*>(a.x + row + k0);
      x[t][4 * q + 0] = xv.x; x[t][4 * q + 1] = xv.y; x[t][4 * q + 2] = xv.z; x[t][4 * q + 3] = xv.w;
    }

  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  int keep = 0;
  const int64_t traj_row = active ? chain * (int64_t)a.n_kept * dim : 0;

  for (int step = 0; step < a.k_steps; ++step) {
    if (a.table) {
      const float4 tb = a.table[step];
      eta = tb.x; sqrt_eta = tb.y; noise_coef = tb.z;
    }
    // ---- g^T = Ps d^T on the matrix cores
    f32x16 g[NT];
    if constexpr (FAST) {
      // The step's normals, drawn in STAGES that the contraction places behind its MFMAs: per quad one stage sets the
      // Philox counter, ten run one round each, two do a Box-Muller pair each (~8 .. 30 VALU instructions a stage).
      uint64_t e_row = (uint64_t)chain * (uint64_t)dim;
      asm volatile("" : "+v"(e_row));
      f32x16 eps[NT];
      constexpr int QUADS = 4 * HIDE, PER_QUAD = 13, STAGES = QUADS * PER_QUAD;
      constexpr int N_MFMA = GKR > 0 ? Mix::kMfmas : 6 * NT * KBU;
      constexpr int PER_MFMA = (STAGES + N_MFMA - 1) / N_MFMA;
      uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, k0 = 0, k1 = 0;
      auto stage = [&](auto sc) {
        constexpr int S = decltype(sc)::value;
        if constexpr (S < STAGES) {
          constexpr int qd = S / PER_QUAD, sub = S % PER_QUAD;
          if constexpr (sub == 0) {
            const uint64_t grp = (e_row + (uint64_t)(32 * (qd >> 2) + 8 * (qd & 3) + 4 * h)) >> 2;
            const uint64_t stp = a.step0 + (uint64_t)step;
            c0 = (uint32_t)grp; c1 = (uint32_t)(grp >> 32); c2 = (uint32_t)stp; c3 = (uint32_t)(stp >> 32);
            k0 = a.key.k0; k1 = a.key.k1;
          } else if constexpr (sub <= 10) {  // one round of philox4x32_10 (ebm_common.h)
            const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
            const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
            const uint32_t n0 = xor3((uint32_t)(p1 >> 32), c1, k0);
            const uint32_t n2 = xor3((uint32_t)(p0 >> 32), c3, k1);
            c1 = (uint32_t)p1; c3 = (uint32_t)p0; c0 = n0; c2 = n2;
            k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
          } else if constexpr (sub == 11) {
            float n0, n1;
            box_muller(c0, c1, n0, n1);
            eps[qd >> 2][4 * (qd & 3) + 0] = n0; eps[qd >> 2][4 * (qd & 3) + 1] = n1;
          } else {
            float n0, n1;
            box_muller(c2, c3, n0, n1);
            eps[qd >> 2][4 * (qd & 3) + 2] = n0; eps[qd >> 2][4 * (qd & 3) + 3] = n1;
          }
        }
      };
      auto behind_mfma = [&](auto ord) {
        gauss3::static_for<PER_MFMA>([&](auto u) { stage(std::integral_constant<int, decltype(ord)::value * PER_MFMA + decltype(u)::value>{}); });
        __builtin_amdgcn_sched_barrier(0);
      };
      if constexpr (GKR > 0) Mix::grad(a.gm, gauss_smem, x, g, lane, behind_mfma);
      else gauss3::contract<NT, KBU>(aop, mus, x, g, lane, behind_mfma);
      // (PER_MFMA * N_MFMA >= STAGES: nothing is left over)
      static_assert(PER_MFMA * N_MFMA >= STAGES, "every stage has an MFMA to hide behind");
      if constexpr (HIDE < NT) {  // the remaining tiles: drawn now, one quad at a time
#pragma unroll
        for (int t = HIDE; t < NT; ++t)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const F4 n4 = normal4_at(a.key, (e_row + (uint64_t)(32 * t + 8 * q + 4 * h)) >> 2, a.step0 + (uint64_t)step);
#pragma unroll
            for (int i = 0; i < 4; ++i) eps[t][4 * q + i] = n4.v[i];
            // update this quad at once: its normals do not stay live
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const int r = 4 * q + i;
              const float x1 = x[t][r] - eta * g[t][r];
              const float dw = eps[t][r] * sqrt_eta;
              x[t][r] = x1 + noise_coef * dw;
            }
          }
      }
#pragma unroll
      for (int t = 0; t < HIDE; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float x1 = x[t][r] - eta * g[t][r];
          const float dw = eps[t][r] * sqrt_eta;
          float nv = x1 + noise_coef * dw;
          if constexpr (GKR > 0) nv = 32 * t + 8 * (r >> 2) + 4 * h < dim ? nv : 0.0f;  // mixture: padding held at 0
          x[t][r] = nv;
        }
    } else {
    if constexpr (GKR > 0) {
      Mix::grad(a.gm, gauss_smem, x, g, lane);
    } else if constexpr (B3) {
      gauss3::contract<NT, KBU>(aop, mus, x, g, lane);  // (contract_pieces costs this body registers: it has no spill to cure)
    } else {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) g[t][r] = 0.0f;
    // software-pipelined: the LDS operands of K-step s+1 are requested before the MFMAs of K-step s
    // issue, so their latency hides under 2*NT*64 matrix-pipe cycles
    auto k_of = [&](int s) { return 32 * (s >> 4) + (s & 3) + 8 * ((s & 15) >> 2) + 4 * h; };
    float pa[NT], pb[NT], ma, mb;
#pragma unroll
    for (int it = 0; it < NT; ++it) pa[it] = Ps[k_of(0) * DIM + 32 * it + m];
    ma = mus[k_of(0)];
    // (Slotting the step's Philox + Box-Muller work between the MFMA groups was tried and buys nothing:
    //  SQ_VALU_MFMA_COEXEC_CYCLES reads 0 for this kernel -- the f32 MFMA executes on the same FP32 lanes
    //  as the VALU, so the two never overlap; profiles/r01_pmc_gauss_mfma.txt.)
#pragma unroll
    for (int s = 0; s < 16 * NT; ++s) {
      if (s + 1 < 16 * NT) {
        const int kn = k_of(s + 1);
#pragma unroll
        for (int it = 0; it < NT; ++it) pb[it] = Ps[kn * DIM + 32 * it + m];
        mb = mus[kn];
      }
      const float d = x[s >> 4][s & 15] - ma;  // B[k = h][m]: the K index this half holds in register s & 15
#pragma unroll
      for (int it = 0; it < NT; ++it)           // A[row = m][k = h] = Ps[32 it + m][k] (symmetric: read as a row)
        g[it] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[it], d, g[it], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int it = 0; it < NT; ++it) pa[it] = pb[it];
      ma = mb;
    }
    }  // exact-f32 MFMA
    // ---- Euler-Maruyama update in the reference's op order, one Philox counter per register quad
    // (the counters are formed here at every step: hoisted out of the step loop they are two registers per quad)
    uint64_t e_row = (uint64_t)chain * (uint64_t)dim;
    asm volatile("" : "+v"(e_row));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k0 = 32 * t + 8 * q + 4 * h;
        F4 eps;
        if (a.noise) {
          float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
          if (active) nv = *reinterpret_cast<const float4*>(a.noise + ((int64_t)step * a.n_chains) * dim + row + k0);
          eps.v[0] = nv.x; eps.v[1] = nv.y; eps.v[2] = nv.z; eps.v[3] = nv.w;
        } else {
          eps = normal4_at(a.key, (e_row + (uint64_t)k0) >> 2, a.step0 + (uint64_t)step);
        }
        // (padding quads run the same straight-line code -- a branch here costs 160 VGPRs -- and whatever
        //  they hold never reaches a real coordinate: their columns of Ps are zero and they are never stored)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float x1 = x[t][4 * q + i] - eta * g[t][4 * q + i];
          const float dw = eps.v[i] * sqrt_eta;
          float nv = x1 + noise_coef * dw;
          if (a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
          // (mixture: padding coordinates are held at 0 -- their "gradient" is x / sigma^2, and a select is free where the
          //  Gaussian's zero rows of Ps make it unnecessary)
          if constexpr (GKR > 0) nv = k0 < dim ? nv : 0.0f;
          x[t][4 * q + i] = nv;
        }
        if constexpr (NT >= 3) __builtin_amdgcn_sched_barrier(0);  // one Philox call's temporaries at a time
      }
    }  // !FAST
    if ((a.traj || DIAG) && --until_keep == 0) {
      until_keep = a.thin;
      if (a.traj && active) {
        if (a.pack > 1) {  // packed rows: element j of the row is coordinate j % sub_dim of chain pack * row + j / sub_dim
          // (sub_dim laundered INSIDE the branch: visible, the 64 per-element divisions and addresses of this path are
          //  loop-invariant, get hoisted into the step loop and spilled there -- 53 scratch stores per lane and step of
          //  every call, packed or not: the dim-128 / 160 kernels wrote 2.7 - 4x their state size per step)
          int sd = a.sub_dim;
          asm volatile("" : "+s"(sd));
          const int64_t kept = keep_off / dim;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
              if (j < dim) {
                const int sub = j / sd;
                a.traj[((chain * a.pack + sub) * (int64_t)a.n_kept + kept) * sd + (j - sub * sd)] = x[t][r];
              }
            }
        } else {
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (32 * t + 8 * q + 4 * h < dim)
                *reinterpret_cast<float4*>(a.traj + traj_row + keep_off + 32 * t + 8 * q + 4 * h) =
                    make_float4(x[t][4 * q], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]);
        }
      }
      keep_off += dim;
      if constexpr (DIAG) {
        // langevin_dynamics.py:170-185: population mean / var per coordinate, mean energy of the kept state -- one record
        // per WAVE of 32 chains straight from the C/D registers (diag::wave_record: no LDS tile, every dim the kernels take)
        const int64_t wave_id = (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
        diag::wave_record<NT>(a.diag.partials, a.diag.n_blocks, keep, wave_id, dim, [&](int t, int r) { return x[t][r]; }, active, lane);
        float e_now;
        if constexpr (GKR > 0) {
          e_now = Mix::energy(a.gm, gauss_smem, x, lane);
        } else {
          f32x16 g2[NT];
          gauss3::contract<NT, KBU>(aop, mus, x, g2, lane);
          float acc = 0.0f;
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float4 mq = *reinterpret_cast<const float4*>(mus + 32 * t + 8 * q + 4 * h);
              acc = __builtin_fmaf(x[t][4 * q] - mq.x, g2[t][4 * q], acc);
              acc = __builtin_fmaf(x[t][4 * q + 1] - mq.y, g2[t][4 * q + 1], acc);
              acc = __builtin_fmaf(x[t][4 * q + 2] - mq.z, g2[t][4 * q + 2], acc);
              acc = __builtin_fmaf(x[t][4 * q + 3] - mq.w, g2[t][4 * q + 3], acc);
            }
          acc += __shfl_xor(acc, 32);
          e_now = 0.5f * acc;
        }
        diag::wave_record_tail(a.diag.partials, a.diag.n_blocks, keep, wave_id, dim, e_now, active, false, lane);
        ++keep;
      }
    }
  }
  if (active) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (32 * t + 8 * q + 4 * h < dim)
          *reinterpret_cast<float4*>(a.x + row + 32 * t + 8 * q + 4 * h) =
              make_float4(x[t][4 * q], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]);
  }
}

// (no minimum-waves bound on the exact-f32 form: at dim 128 the state alone is 128 VGPRs and that kernel needs 432 -- one
//  wave per SIMD without spills was 4.6 ms on 2^18 x 128 x 50 where a 256-VGPR cap with spills was 6.9 ms)
template <int NT>
__global__ __launch_bounds__(kBlock) void gauss_langevin_mfma_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, false>(a);
}
template <int NT, int KT = 0>
__global__ __launch_bounds__(kBlock) void gauss_langevin_bf16x3_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, false, kBlock, NT, 0, false, KT>(a);
}
template <int NT, int KT = 0>
__global__ __launch_bounds__(kBlock) void gauss_langevin_bf16x3_fast_kernel(GaussArgs a) {
  // five tiles: the normals of four are drawn behind the MFMAs, the fifth tile's after the contraction -- all five (80
  // registers across the contraction) left the 512-register wave with 76 spilled values
  gauss_langevin_mfma_body<NT, true, true, kBlock, (NT >= 5 ? 4 : NT), 0, false, KT>(a);
}
template <int NT, int GKR>
__global__ __launch_bounds__(kBlock) void gmm_langevin_bf16x3_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, false, kBlock, NT, GKR>(a);
}
template <int NT, int GKR>
__global__ __launch_bounds__(kBlock) void gmm_langevin_bf16x3_fast_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, true, kBlock, NT, GKR>(a);
}
// with diagnostics records (GKR = 0: the dense Gaussian)
template <int NT, int GKR>
__global__ __launch_bounds__(kBlock) void matrix_langevin_diag_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, false, kBlock, NT, GKR, true>(a);
}
constexpr int kWideBlock = 512;
template <int NT, int HIDE, int KT = 0>
__global__ __launch_bounds__(kWideBlock) void gauss_langevin_bf16x3_fast_wide_kernel(GaussArgs a) {
  gauss_langevin_mfma_body<NT, true, true, kWideBlock, HIDE, 0, false, KT>(a);
}

template <int NT, int KT = 0>
int launch_nt(const GaussArgs& a, hipStream_t st) {
  // A/B switch for tests and profiling: EBM_GAUSS_F32MFMA=1 keeps the exact-f32 MFMA contraction
  static const bool f32_mfma = ab_switch("EBM_GAUSS_F32MFMA");
  const size_t smem = (f32_mfma ? (size_t)(32 * NT) * (32 * NT) * sizeof(float) : gauss3::aop_bytes(NT)) + 32 * NT * sizeof(float);
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  if (attr_once.first() && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_langevin_mfma_kernel<NT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_langevin_bf16x3_kernel<NT, KT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_langevin_bf16x3_fast_kernel<NT, KT>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int64_t blocks = ceil_div64(a.n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  if constexpr (NT == 3) {  // (four tiles: the 256-register cap costs 80 B of scratch and the unhidden half of the RNG -- no gain)
    // A/B switch: EBM_GAUSS_WIDE=0 keeps the 256-thread workgroups (dim 96: 1.81 ms against 1.68)
    static const bool wide_off = ab_switch("EBM_GAUSS_WIDE", '0');
    if (!f32_mfma && !a.noise && !a.clamp_on && !wide_off) {
      constexpr int HIDE = 2;
      static bool wide_attr = false;
      if (!wide_attr && smem > 64 * 1024) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gauss_langevin_bf16x3_fast_wide_kernel<NT, HIDE, KT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        wide_attr = true;
      }
      const int64_t wblocks = ceil_div64(a.n_chains, 32 * (kWideBlock / 64));
      hipLaunchKernelGGL((gauss_langevin_bf16x3_fast_wide_kernel<NT, HIDE, KT>), dim3((unsigned)wblocks), dim3(kWideBlock), smem, st, a);
      return check_launch("ebm_langevin_chain_f32");
    }
  }
  if (f32_mfma) hipLaunchKernelGGL(gauss_langevin_mfma_kernel<NT>, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else if (!a.noise && !a.clamp_on) hipLaunchKernelGGL((gauss_langevin_bf16x3_fast_kernel<NT, KT>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else hipLaunchKernelGGL((gauss_langevin_bf16x3_kernel<NT, KT>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_langevin_chain_f32");
}

}  // namespace

// dims that are multiples of 4 run on zero-padded 32-wide tiles; below 20 the padding waste outweighs the
// matrix cores (measured: scripts/bench_gauss_dims.py), those stay on the lane-group kernel
bool gauss_mfma_supported(int32_t dim) { return dim >= 20 && dim <= 128 && (dim % 4) == 0; }
// the plain Langevin call only (no records, no HMC): five tiles, 150 KB of split operands + mu in the CU's 160 KB
bool gauss_lds5_supported(int32_t dim) { return dim > 128 && dim <= 160 && (dim % 4) == 0; }

// Other widths (below 20, or not a multiple of 4) PACKED: `pack` consecutive chains of the row-major state are one row of
// width pack * dim, whose Gaussian is block diagonal -- kron(I, Ps), the mean repeated.  The flat element order, hence the
// Philox field and the update of every element, is unchanged; the kernel only stages the block-diagonal matrix (and scatters
// trajectory rows).  1: no packing needed; 0: no packing possible (n not divisible, or no factor lands in 20 .. 128, % 4).
// Langevin only: an HMC accept decision is per chain.  dim 2 keeps its two-chains-per-lane kernel.
int32_t gauss_pack_factor(int32_t dim, int64_t n_chains) {
  if (gauss_mfma_supported(dim)) return 1;
  if (dim < 3) return 0;
  for (int32_t g = 2; g <= 16; g *= 2) {
    const int32_t d = g * dim;
    if (d > 128) break;
    if (d >= 20 && d % 4 == 0 && n_chains % g == 0) return g;
  }
  return 0;
}

int launch_langevin_chain_gauss_mfma(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                                     float eta, float sqrt_eta, float noise_coef, const float* coef_table,
                                     int clamp_on, float cmin, float cmax, int32_t thin, float* traj,
                                     const float* noise, uint64_t seed, uint64_t offset, hipStream_t st) {
  GaussArgs a{};
  // (dims 132 .. 160: FIVE tiles -- the three splits of Ps are 150 KB, the last width whose precision matrix stays resident in LDS)
  const int32_t pack = gauss_lds5_supported(dim) ? 1 : gauss_pack_factor(dim, n_chains);
  if (pack < 1) return fail(EBM_EDIM, "ebm_langevin_chain_f32: no matrix-layout form for a Gaussian of dim %d over %lld chains", dim, (long long)n_chains);
  a.sub_dim = dim; a.pack = pack;
  n_chains /= pack; dim *= pack;  // the packed geometry from here on
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.mean = e.dev0; a.prec = e.dev1;
  a.gm = gmm3::Params{nullptr, nullptr, 0, dim, 0.0f, 0.0f};
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0}; a.diag_offset_floats = 0;
  // the last 16 coordinates of the last tile all padding: that K-block is left out (EBM_GAUSS_NOTRIM=1: the A/B switch)
  static const bool no_trim = ab_switch("EBM_GAUSS_NOTRIM");
  const int nt = (dim + 31) / 32;
  const bool trim = 32 * nt - dim >= 16 && !no_trim;
  switch (nt) {
    case 1: return launch_nt<1>(a, st);
    case 2: return trim ? launch_nt<2, 1>(a, st) : launch_nt<2>(a, st);
    case 3: return trim ? launch_nt<3, 1>(a, st) : launch_nt<3>(a, st);
    case 4: return trim ? launch_nt<4, 1>(a, st) : launch_nt<4>(a, st);
    default: return trim ? launch_nt<5, 1>(a, st) : launch_nt<5>(a, st);
  }
}

// ---------------------------------------------------------------------------------
// Gaussian mixture Langevin chain on the matrix layout
// ---------------------------------------------------------------------------------
bool gmm_mfma_supported(int32_t dim, int32_t n_comp) { return dim >= 20 && dim <= 128 && (dim % 4) == 0 && n_comp >= 1 && n_comp <= 32; }

namespace {
template <int NT, int GKR>
int launch_gmm_langevin(const GaussArgs& a, hipStream_t st) {
  const size_t smem = (size_t)gmm3::Mixture<NT, GKR>::kLdsFloats * sizeof(float);
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  if (attr_once.first() && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_langevin_bf16x3_kernel<NT, GKR>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gmm_langevin_bf16x3_fast_kernel<NT, GKR>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int64_t blocks = ceil_div64(a.n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  // no clamp, no injected noise: the step's normals are drawn in stages behind the MFMAs (as for the Gaussian)
  static const bool no_fast = ab_switch("EBM_GMM_NOFAST");
  // (three tiles: the staged form drops to one wave per SIMD -- 1.54 ms against 1.39 at dim 96 -- and stays off)
  if (NT != 3 && !a.noise && !a.clamp_on && !no_fast)
    hipLaunchKernelGGL((gmm_langevin_bf16x3_fast_kernel<NT, GKR>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  else
    hipLaunchKernelGGL((gmm_langevin_bf16x3_kernel<NT, GKR>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_langevin_chain_f32");
}
template <int NT>
int launch_gmm_langevin_nt(const GaussArgs& a, hipStream_t st) {
  if (a.gm.n_comp <= 8) return launch_gmm_langevin<NT, 4>(a, st);
  if (a.gm.n_comp <= 16) return launch_gmm_langevin<NT, 8>(a, st);
  return launch_gmm_langevin<NT, 16>(a, st);
}
}  // namespace

int launch_langevin_chain_gmm_mfma(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                                   float eta, float sqrt_eta, float noise_coef, const float* coef_table,
                                   int clamp_on, float cmin, float cmax, int32_t thin, float* traj,
                                   const float* noise, uint64_t seed, uint64_t offset, hipStream_t st) {
  GaussArgs a{};
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.mean = nullptr; a.prec = nullptr;
  a.sub_dim = dim; a.pack = 1;  // (mixtures do not pack)
  a.gm = gmm3::Params{e.dev0, e.dev1, e.n_comp, dim, e.s[0], e.s[1]};
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0}; a.diag_offset_floats = 0;
  switch ((dim + 31) / 32) {
    case 1: return launch_gmm_langevin_nt<1>(a, st);
    case 2: return launch_gmm_langevin_nt<2>(a, st);
    case 3: return launch_gmm_langevin_nt<3>(a, st);
    default: return launch_gmm_langevin_nt<4>(a, st);
  }
}

// ---------------------------------------------------------------------------------
// Diagnostics records on the matrix-layout Langevin kernels (dense Gaussian, mixtures): every dim they take -- one record per
// wave of 32 chains from the C/D registers (round 3; rounds 1-2 went through an LDS tile of the workgroup's chains, which did
// not fit beyond dim 96: a call WITH records then ran on another kernel family than the same call without).
// ---------------------------------------------------------------------------------
bool matrix_langevin_diag_plan(const ebm_energy_t& e, int64_t n_chains, int32_t dim, diag::DiagArgs& d) {
  const int32_t pack = e.kind == EBM_ENERGY_GAUSSIAN ? gauss_pack_factor(dim, n_chains) : 0;
  const bool mix = e.kind == EBM_ENERGY_GMM && gmm_mfma_supported(dim, e.n_comp) && !(dim == 32 && e.n_comp <= 8);
  if (!(pack >= 1 || mix)) return false;
  if (pack > 1)  // packed rows: the records are those of n / pack rows of width pack * dim (S > dim tells the caller to fold)
    return diag::plan(n_chains / pack, pack * dim, 32 * (int64_t)pack * dim, d);
  return diag::plan(n_chains, dim, 32 * (int64_t)dim, d);
}

namespace {
template <int NT, int GKR>
int launch_matrix_diag(GaussArgs& a, hipStream_t st) {
  const size_t energy_floats = GKR > 0 ? (size_t)gmm3::Mixture<NT, GKR == 0 ? 4 : GKR>::kLdsFloats
                                       : gauss3::aop_bytes(NT) / sizeof(float) + 32 * NT;
  const size_t smem = energy_floats * sizeof(float);
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  if (attr_once.first() && smem > 64 * 1024) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(matrix_langevin_diag_kernel<NT, GKR>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int64_t blocks = ceil_div64(a.n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  hipLaunchKernelGGL((matrix_langevin_diag_kernel<NT, GKR>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_langevin_chain_f32");
}
template <int NT>
int launch_matrix_diag_nt(GaussArgs& a, bool mixture, hipStream_t st) {
  if (!mixture) return launch_matrix_diag<NT, 0>(a, st);
  if (a.gm.n_comp <= 8) return launch_matrix_diag<NT, 4>(a, st);
  if (a.gm.n_comp <= 16) return launch_matrix_diag<NT, 8>(a, st);
  return launch_matrix_diag<NT, 16>(a, st);
}
}  // namespace

int launch_langevin_chain_matrix_diag(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                                      float eta, float sqrt_eta, float noise_coef, const float* coef_table,
                                      int clamp_on, float cmin, float cmax, int32_t thin, float* traj,
                                      const float* noise, uint64_t seed, uint64_t offset, float* diag_partials, hipStream_t st) {
  GaussArgs a{};
  if (!matrix_langevin_diag_plan(e, n_chains, dim, a.diag))
    return fail(EBM_EDIM, "ebm_langevin_chain_f32: no matrix-layout diagnostics records for this energy / dim %d", dim);
  const bool mixture = e.kind == EBM_ENERGY_GMM;
  const int32_t pack = mixture ? 1 : gauss_pack_factor(dim, n_chains);
  a.sub_dim = dim; a.pack = pack;
  n_chains /= pack; dim *= pack;  // the packed geometry from here on
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  a.mean = mixture ? nullptr : e.dev0; a.prec = mixture ? nullptr : e.dev1;
  a.gm = mixture ? gmm3::Params{e.dev0, e.dev1, e.n_comp, dim, e.s[0], e.s[1]} : gmm3::Params{nullptr, nullptr, 0, dim, 0.0f, 0.0f};
  a.diag.partials = diag_partials;
  switch ((dim + 31) / 32) {
    case 1: return launch_matrix_diag_nt<1>(a, mixture, st);
    case 2: return launch_matrix_diag_nt<2>(a, mixture, st);
    case 3: return launch_matrix_diag_nt<3>(a, mixture, st);
    default: return launch_matrix_diag_nt<4>(a, mixture, st);
  }
}

}  // namespace ebm
