// (shared by gauss_hmc_stream.hip and matrix_hmc_diag.hip)
// HMC transitions for the dense Gaussian energy at dims 164 .. 256 (multiples of 4): the matrix-layout transition body
// (mfma_hmc_body.h: position, momentum and force in the C/D layout of 32 x 32 tiles, one wave per 32 chains) with an
// evaluation that STREAMS the precision matrix -- the three bf16 images of a stage (32 columns of Ps for all 32 NT rows)
// arrive ready-made from the pre-split copy of Ps (ebm_gauss_prec_image_f32: the resident Langevin kernel's layout,
// gauss_big_body.h) by LDS-direct loads, double-buffered, one stage ahead of the MFMAs that read them; the B operands are split
// from the position registers in slots behind the MFMAs.  Up to 160 dims the images stay resident in LDS (gauss_hmc_mfma.hip);
// beyond 256 the position, momentum and force of 32 chains no longer fit a wave's registers (the sampler's GEMM route).
// Before round 4 these widths ran per transition as library GEMMs + element-wise kernels: dim 256, 2^17 chains, 5 transitions
// of 10 leapfrog steps 18.6 ms.
// Reference: samplers/hmc.py:201-315 (transition), integrators/leapfrog.py:116-187, core/base_model.py:181-210 (energy).
#pragma once
#include "mfma_hmc_body.h"
#include "gauss_big_body.h"

namespace ebm {
namespace {

template <int NT>
struct GaussStreamE {
  using C = gbig::ResCfg<NT>;
  static constexpr int SLABU = C::SLABU;
  static constexpr uint32_t STAGE_BYTES = 3u * SLABU * 16u;
  static constexpr int kSlabFloats = (int)(2u * STAGE_BYTES / 4u);
  static constexpr int kLdsFloats = kSlabFloats + 32 * NT;  // [2 buffers][3 pieces][SLABU units] | mu
  static constexpr bool kEvalGivesEnergy = true;
  static constexpr bool kCarry = false;      // (no LDS left for a parked force: L + 1 evaluations per transition, as the reference)
  static constexpr bool kBlockVote = true;   // barriers inside eval()
  int gstage = 0;  // stages done so far: its parity is the buffer the next stage reads (NT may be odd; the pipeline runs across calls)

  // request stage s of the image into buffer `buf`: 6 NT pieces of 1 KiB dealt round-robin to the four waves (assembly: see
  // gauss_big_body.h -- behind the builtin the compiler serialises every later LDS read)
  __device__ static __forceinline__ void dma(const GaussHmcArgs& a, const float* lds, int buf, int s) {
    const char* src = a.prec_image + gbig::big_image_bytes<NT, 1>(32 * NT) + (size_t)s * STAGE_BYTES;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const uint32_t dst = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const float*)lds + (uint32_t)buf * STAGE_BYTES;
    for (int piece = wv; piece < (int)(STAGE_BYTES / 1024u); piece += kBlock / 64) {
      const uint32_t voff = (uint32_t)(piece * 1024 + lane * 16), base = dst + (uint32_t)piece * 1024u;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(voff), "s"(src), "s"(base) : "memory");
    }
  }
  __device__ static __forceinline__ void stage(const GaussHmcArgs& a, float* lds, int lo) {  // shifted rows (a.prec_image: this class's)
    for (int i = threadIdx.x; i < 32 * NT; i += kBlock) lds[kSlabFloats + i] = (i >= lo && i - lo < a.dim) ? a.mean[i - lo] : 0.0f;
    dma(a, lds, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __device__ static __forceinline__ void stage(const GaussHmcArgs& a, float* lds) {
    for (int i = threadIdx.x; i < 32 * NT; i += kBlock) lds[kSlabFloats + i] = i < a.dim ? a.mean[i] : 0.0f;
    dma(a, lds, 0, 0);  // the first evaluation's first stage (the body's barrier follows)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __device__ static __forceinline__ float energy(const GaussHmcArgs&, const float*, const Tile<NT>&, int, int) { return 0.0f; }

  // g^T = Ps (x - mu)^T, E = 0.5 (x - mu) . g.  Stage s = the 32 columns 32 s .. of Ps = the two K-blocks whose B operands are
  // registers 0 .. 7 and 8 .. 15 of position tile s; six products per (out tile, K-block) as in gauss_bf16x3.h, smallest first.
  __device__ __forceinline__ float eval(const GaussHmcArgs& a, const float* lds, const Tile<NT>& x, Tile<NT>& g, int m, int h) {
    using gbig::SplitJob;
    using gbig::Tri;
    using gauss3::bf16x8;
    using gauss3::static_for;
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const bf16x8* slab = reinterpret_cast<const bf16x8*>(lds);
    const float* mus = lds + kSlabFloats;
    int hs = h, ms = m;
    asm volatile("" : "+v"(hs), "+v"(ms));  // (per call: nothing derived from the lane is hoisted out of the trajectory loop and spilled)
    const int rd_unit[2] = {hs * 32 + ((ms + 2 * hs) & 31), 64 + hs * 32 + ((ms + 2 * (2 + hs)) & 31)};  // this lane's operand slot per K-block
    static_for<NT>([&](auto tc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) g.t[decltype(tc)::value][r] = 0.0f;
    });
    auto b_init = [&](SplitJob& jb, auto tc, auto kc, auto hc) {  // half hc of the eight differences of K-block (t, kb2)
      constexpr int t = decltype(tc)::value, kb2 = decltype(kc)::value, hf = decltype(hc)::value;
      const f32x4 mm = *reinterpret_cast<const f32x4*>(mus + 32 * t + 16 * kb2 + 8 * hf + 4 * hs);
#pragma unroll
      for (int j = 0; j < 4; ++j) jb.d[4 * hf + j] = x.t[t][8 * kb2 + 4 * hf + j] - mm[j];
    };
    Tri b0;
    {
      SplitJob j0;
      b_init(j0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{});
      b_init(j0, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 1>{});
      static_for<8>([&](auto kc) { j0.step(kc); });
      b0 = j0.tri();
    }
    static_for<NT>([&](auto sc) {
      constexpr int s = decltype(sc)::value, sn = (s + 1) % NT;  // behind the last stage: stage 0 of the next evaluation
      constexpr int HALF = 6 * NT, B_STEPS = 10;
      static_assert(2 * B_STEPS <= HALF, "the split work of a stage fits behind its MFMAs");
      const int buf = gstage & 1;
      dma(a, lds, buf ^ 1, sn);  // (the barrier that ended the stage before freed that buffer)
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8* sb = slab + (size_t)buf * 3 * SLABU;
      SplitJob jb1, jb0n;
      auto b_job = [&](SplitJob& jb, auto tc, auto kc, auto kk) {  // step kk of 10 of a B operand
        constexpr int k = decltype(kk)::value;
        if constexpr (k < 2) b_init(jb, tc, kc, std::integral_constant<int, k>{});
        else jb.step(std::integral_constant<int, k - 2>{});
      };
      auto slot = [&](auto oc) {  // behind MFMA o of the stage: the B operands of K-block 1 and of the next stage's K-block 0
        constexpr int o = decltype(oc)::value;
        if constexpr (o < HALF && o / 2 < B_STEPS) {
          if constexpr (o % 2 == 0) b_job(jb1, sc, std::integral_constant<int, 1>{}, std::integral_constant<int, o / 2>{});
          else if constexpr (s + 1 < NT)
            b_job(jb0n, std::integral_constant<int, (s + 1 < NT ? s + 1 : 0)>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, o / 2>{});
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      static_for<2>([&](auto kc) {
        constexpr int kb2 = decltype(kc)::value;
        constexpr int PAIRS = (NT + 1) / 2;
        const Tri bb = kb2 == 0 ? b0 : jb1.tri();
        auto read_a = [&](auto pc, bf16x8 (&a6)[6]) {
          constexpr int pi = decltype(pc)::value, ot0 = 2 * pi, ot1 = 2 * pi + 1 < NT ? 2 * pi + 1 : 2 * pi;
          const bf16x8* sr = sb + rd_unit[kb2];
          a6[0] = sr[2 * SLABU + ot0 * 128]; a6[1] = sr[SLABU + ot0 * 128]; a6[2] = sr[ot0 * 128];
          if constexpr (ot1 != ot0) {
            a6[3] = sr[2 * SLABU + ot1 * 128]; a6[4] = sr[SLABU + ot1 * 128]; a6[5] = sr[ot1 * 128];
          }
        };
        bf16x8 acur[6];
        read_a(std::integral_constant<int, 0>{}, acur);
        static_for<PAIRS>([&](auto pc) {
          constexpr int pi = decltype(pc)::value, ot0 = 2 * pi, ot1 = 2 * pi + 1 < NT ? 2 * pi + 1 : 2 * pi;
          constexpr bool two = ot1 != ot0;
          constexpr int o0 = kb2 * HALF + 12 * pi;  // ordinal of this pair's first MFMA
          bf16x8 anext[6];
          if constexpr (pi + 1 < PAIRS) read_a(std::integral_constant<int, pi + 1>{}, anext);
          __builtin_amdgcn_sched_barrier(0);
          f32x16 g0 = g.t[ot0], g1;
          if constexpr (two) g1 = g.t[ot1];
          static_for<6>([&](auto tc) {  // (term, operand) in issue order: Pl dh | Pm dm, Pm dh | Ph dl, Ph dm, Ph dh
            constexpr int term = decltype(tc)::value;
            constexpr int ai = term == 0 ? 0 : (term <= 2 ? 1 : 2);
            const bf16x8& bp = (term == 0 || term == 2 || term == 5) ? bb.h : ((term == 1 || term == 4) ? bb.m : bb.l);
            g0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[ai], bp, g0, 0, 0, 0);
            slot(std::integral_constant<int, o0 + (two ? 2 : 1) * term>{});
            if constexpr (two) {
              g1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(acur[3 + ai], bp, g1, 0, 0, 0);
              slot(std::integral_constant<int, o0 + 2 * term + 1>{});
            }
          });
          g.t[ot0] = g0;
          if constexpr (two) g.t[ot1] = g1;
          if constexpr (pi + 1 < PAIRS) {
#pragma unroll
            for (int i = 0; i < 6; ++i) acur[i] = anext[i];
          }
        });
      });
      if constexpr (s + 1 < NT) b0 = jb0n.tri();
      ++gstage;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's pieces of the next slab have landed
      __syncthreads();                                    // ... everybody's have, and this slab is read by everyone
    });
    float acc = 0.0f;
    static_for<NT * 4>([&](auto ic) {
      constexpr int t = decltype(ic)::value >> 2, q = decltype(ic)::value & 3;
      const f32x4 mq = *reinterpret_cast<const f32x4*>(mus + 32 * t + 8 * q + 4 * hs);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc = __builtin_fmaf(x.t[t][4 * q + i] - mq[i], g.t[t][4 * q + i], acc);
    });
    acc += __shfl_xor(acc, 32);
    return 0.5f * acc;
  }
};

}  // namespace
}  // namespace ebm
