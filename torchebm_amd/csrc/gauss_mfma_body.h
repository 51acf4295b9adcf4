// The matrix-layout Langevin chain body for the dense Gaussian / isotropic mixtures (kernels and launchers: gauss_mfma.hip,
// gauss_shift.hip).  Split out so that the shifted-row instantiations compile as a translation unit of their own.
#pragma once
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
  int32_t sh_classes = 1;  // SHIFTED rows (SH kernels): 4 / gcd(dim, 4) alignment classes of chains, one per workgroup
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
// SH: SHIFTED rows -- widths that are not a multiple of 4 (21 .. 157).  A row of such a state starts at flat element dim * c,
// off the float4 / Philox-counter grid by o = (dim * c) & 3.  The tile is laid over the ALIGNED flat range that contains the
// row: tile coordinate j = coordinate j - o of the chain, so that a register quad is still one aligned float4 and one Philox
// counter of the flat field (the same numbers the flat kernels draw); the o leading and the trailing tile coordinates belong
// to the neighbouring chains and are padding here (zero rows / columns of the staged matrix, loaded as 0, never stored).
// o depends on c mod 4 only, and the staged matrix on o: a workgroup takes chains of ONE class c = S j + s (S = 4 or 2
// classes, s = blockIdx % S) and stages Ps shifted by its own o.  Before, these widths ran packed (block-diagonal rows of
// 2 - 4 chains: 2 - 4x the matrix work) or, beyond 64, on the lane-group kernel (dim 66 .. 126: 5 - 170 ms where dim 128
// takes 3).
template <int NT, bool B3, bool FAST = false, int BLOCK = 256, int HIDE = NT, int GKR = 0, bool DIAG = false, int KT = 0, bool SH = false>
__device__ __forceinline__ void gauss_langevin_mfma_body(const GaussArgs& a) {
  constexpr int DIM = 32 * NT, KBU = 2 * NT - KT;
  static_assert(KT == 0 || (B3 && GKR == 0), "trimmed K-blocks: the dense Gaussian on the bf16 pipe");
  static_assert(!SH || B3, "shifted rows: the contraction on the bf16 pipe");
  const int sh_s = SH ? (int)(blockIdx.x % (unsigned)a.sh_classes) : 0;
  const int lo = SH ? ((a.dim * sh_s) & 3) : 0;  // first tile coordinate of the row (0 unless SH)
  using Mix = gmm3::Mixture<NT, GKR == 0 ? 4 : GKR>;
  // LDS: the precision matrix -- fp32 [DIM][DIM], or its three operand-ready bf16 splits (1.5x the bytes) -- then mu
  float* Ps = gauss_smem;
  __bf16* aop = reinterpret_cast<__bf16*>(gauss_smem);
  float* mus = gauss_smem + (B3 ? (int)(gauss3::aop_bytes(NT) / sizeof(float)) : DIM * DIM);  // [DIM]
  // dim <= DIM, dim % 4 == 0: the tiles are zero-padded -- padded coordinates stay exactly 0 (d = 0, g = 0,
  // no noise) and whole register quads beyond dim are never loaded, drawn or stored
  const int dim = a.dim;
  gmm3::Params gm = a.gm;  // (the mixture kernels: with this workgroup's row offset)
  gm.lo = lo;
  if constexpr (GKR > 0) {
    Mix::stage(gm, gauss_smem, BLOCK);
  } else {
    // the precision of the (possibly packed) row: block-diagonal copies of the sub_dim x sub_dim matrix
    const int sd = a.sub_dim;
    const auto ps_at = [&](int r, int c) {
      r -= lo; c -= lo;
      if (r < 0 || c < 0 || r >= dim || c >= dim) return 0.0f;
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
    for (int i = threadIdx.x; i < DIM; i += BLOCK) mus[i] = (i >= lo && i - lo < dim) ? a.mean[(i - lo) % a.sub_dim] : 0.0f;
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int m = lane & 31, h = lane >> 5;
  const int64_t chain = SH ? (((int64_t)(blockIdx.x / (unsigned)a.sh_classes) * (BLOCK / 64) + (threadIdx.x >> 6)) * 32 + m) * a.sh_classes + sh_s
                           : ((int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6)) * 32 + m;
  const bool active = chain < a.n_chains;
  const int hi = lo + dim;  // one past the row's last tile coordinate
  const int64_t row = active ? chain * (int64_t)dim - lo : 0;  // flat element of tile coordinate 0 (SH: a multiple of 4 all the same)

  // state in the C/D layout; quad q of tile t = coordinates 32t + 8q + 4h .. +3
  f32x16 x[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k0 = 32 * t + 8 * q + 4 * h;
      float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);
      if constexpr (SH) {
        if (active && k0 >= lo && k0 + 3 < hi) {
          xv = *reinterpret_cast<const float4*>(a.x + row + k0);
        } else if (active && k0 + 3 >= lo && k0 < hi) {  // a quad shared with a neighbouring chain: its own elements only
          if (k0 + 0 >= lo && k0 + 0 < hi) xv.x = a.x[row + k0 + 0];
          if (k0 + 1 >= lo && k0 + 1 < hi) xv.y = a.x[row + k0 + 1];
          if (k0 + 2 >= lo && k0 + 2 < hi) xv.z = a.x[row + k0 + 2];
          if (k0 + 3 >= lo && k0 + 3 < hi) xv.w = a.x[row + k0 + 3];
        }
      } else {
        if (active && k0 < dim) xv = *reinterpret_cast<const float4*>(a.x + row + k0);
      }
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
      uint64_t e_row = (uint64_t)chain * (uint64_t)dim - (uint64_t)lo;
      asm volatile("" : "+v"(e_row));
      f32x16 eps[NT];
      // (round 6) 21 stages per quad of <= 5 instructions -- counter | ten Philox rounds | two Box-Muller pairs in five stages each --
      // dealt EVENLY over the contraction's MFMAs (stages [S ord / N, S (ord + 1) / N) behind MFMA ord).  Before: 13 stages of 6 - 20
      // instructions, two behind each of the first STAGES / 2 MFMAs and none behind the rest (dim 128: 104 gaps of 8 - 17
      // instructions, 88 empty ones -- scripts/isa_gapmap.py); five plain instructions hide behind a 32-cycle MFMA, every further
      // one costs its issue time, an empty gap wastes it (profiles/r06_mfma_valu_overlap.txt).
      constexpr int QUADS = 4 * HIDE, PER_QUAD = 21, STAGES = QUADS * PER_QUAD;
      constexpr int N_MFMA = GKR > 0 ? Mix::kMfmas : 6 * NT * KBU;
      uint32_t c0 = 0, c1 = 0, c2 = 0, c3 = 0, k0 = 0, k1 = 0;
      float bu = 0.0f, brev = 0.0f, br = 0.0f, bs = 0.0f;
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
          } else {  // box_muller (ebm_common.h) on (c0, c1), then on (c2, c3): the same operations, five stages each
            constexpr int pr = (sub - 11) / 5, st = (sub - 11) % 5;
            if constexpr (st == 0) {
              bu = u01_open_low(pr == 0 ? c0 : c2);
              brev = (float)(pr == 0 ? c1 : c3) * 0x1p-32f;
            } else if constexpr (st == 1) {
              br = -1.38629436111989061883f * __builtin_amdgcn_logf(bu);
            } else if constexpr (st == 2) {
              br = __builtin_amdgcn_sqrtf(br);
              bs = __builtin_amdgcn_sinf(brev);
            } else if constexpr (st == 3) {
              brev = __builtin_amdgcn_cosf(brev);
            } else {
              eps[qd >> 2][4 * (qd & 3) + 2 * pr] = br * bs;
              eps[qd >> 2][4 * (qd & 3) + 2 * pr + 1] = br * brev;
            }
          }
        }
      };
      auto behind_mfma = [&](auto ord) {
        constexpr int o = decltype(ord)::value, s0 = (int)((long long)STAGES * o / N_MFMA), s1 = (int)((long long)STAGES * (o + 1) / N_MFMA);
        gauss3::static_for<s1 - s0>([&](auto u) { stage(std::integral_constant<int, s0 + decltype(u)::value>{}); });
        __builtin_amdgcn_sched_barrier(0);
      };
      if constexpr (GKR > 0) Mix::grad(gm, gauss_smem, x, g, lane, behind_mfma);
      else gauss3::contract<NT, KBU>(aop, mus, x, g, lane, behind_mfma);
      // (the last MFMA's share ends at STAGES: nothing is left over)
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
          if constexpr (GKR > 0 && SH) {
            const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
            nv = (j >= lo && j < hi) ? nv : 0.0f;
          } else if constexpr (GKR > 0) nv = 32 * t + 8 * (r >> 2) + 4 * h < dim ? nv : 0.0f;  // mixture: padding held at 0
          x[t][r] = nv;
        }
    } else {
    if constexpr (GKR > 0) {
      Mix::grad(gm, gauss_smem, x, g, lane);
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
    uint64_t e_row = (uint64_t)chain * (uint64_t)dim - (uint64_t)lo;
    asm volatile("" : "+v"(e_row));
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k0 = 32 * t + 8 * q + 4 * h;
        F4 eps;
        if (a.noise) {
          float4 nv = make_float4(0.f, 0.f, 0.f, 0.f);
          if constexpr (SH) {  // (an injected field restarts at step * n * dim: no alignment to count on)
            const float* nz = a.noise + ((int64_t)step * a.n_chains) * dim + row + k0;
            if (active && k0 + 0 >= lo && k0 + 0 < hi) nv.x = nz[0];
            if (active && k0 + 1 >= lo && k0 + 1 < hi) nv.y = nz[1];
            if (active && k0 + 2 >= lo && k0 + 2 < hi) nv.z = nz[2];
            if (active && k0 + 3 >= lo && k0 + 3 < hi) nv.w = nz[3];
          } else
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
          if constexpr (GKR > 0 && SH) nv = (k0 + i >= lo && k0 + i < hi) ? nv : 0.0f;
          else if constexpr (GKR > 0) nv = k0 < dim ? nv : 0.0f;
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
        } else if constexpr (SH) {  // (a kept row starts wherever (chain * n_kept + kept) * dim falls)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * h;
              if (j >= lo && j < hi) a.traj[traj_row + keep_off + (j - lo)] = x[t][r];
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
        // (SH: the records of the K classes interleave -- record (group, class), diag.h plan_classes)
        const int64_t wave_id = SH ? ((int64_t)(blockIdx.x / (unsigned)a.sh_classes) * (BLOCK / 64) + (threadIdx.x >> 6)) * a.sh_classes + sh_s
                                   : (int64_t)blockIdx.x * (BLOCK / 64) + (threadIdx.x >> 6);
        diag::wave_record<NT>(a.diag.partials, a.diag.n_blocks, keep, wave_id, dim, [&](int t, int r) { return x[t][r]; }, active, lane, 0, lo);
        float e_now;
        if constexpr (GKR > 0) {
          e_now = Mix::energy(gm, gauss_smem, x, lane);
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
      for (int q = 0; q < 4; ++q) {
        const int k0 = 32 * t + 8 * q + 4 * h;
        if constexpr (SH) {
          if (k0 >= lo && k0 + 3 < hi) {
            *reinterpret_cast<float4*>(a.x + row + k0) = make_float4(x[t][4 * q], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]);
          } else if (k0 + 3 >= lo && k0 < hi) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
              if (k0 + i >= lo && k0 + i < hi) a.x[row + k0 + i] = x[t][4 * q + i];
          }
        } else {
          if (k0 < dim)
            *reinterpret_cast<float4*>(a.x + row + k0) = make_float4(x[t][4 * q], x[t][4 * q + 1], x[t][4 * q + 2], x[t][4 * q + 3]);
        }
      }
  }
}


}  // namespace
}  // namespace ebm
