// Kernel body of the wide-MLP HMC transition kernel: see mlp_wide_hmc.hip.  Four translation units instantiate it (compiled in
// parallel): mlp_wide_hmc.hip / mlp_wide_hmc_diag.hip (H = 64, 128; scalar or identity / diagonal mass) and
// mlp_stream_hmc.hip / mlp_stream_hmc_diag.hip (H = 256).
#pragma once
#include "mlp_wide_body.h"

namespace ebm {
namespace widemlp {

struct WideHmcArgs {
  float* x;
  int64_t n_chains;
  int32_t dim, n_mh, n_leapfrog;
  float eps;
  const float* eps_table;
  int32_t mass_kind;
  float mass_raw, mass_sqrt, mass_safe;
  const float* mass_diag;
  int32_t thin, n_kept;
  float* traj;
  uint8_t* accept_mask;
  uint32_t* accept_count;
  const float* p_noise;
  const float* u;
  RngKey key;
  uint64_t step0;
  const float* params;
  float* diag_partials;  // in-kernel diagnostics records (mlp_wide_body.h: one per wave), or null
  int64_t diag_blocks;
  const char* w1_image;  // MODE 3 (mlp_wide_body.h), or null
};

// FAST: the plain call only -- in-kernel momenta and uniforms, no diagnostics records, dim % 4 == 0 or dim == 2 -- with the
// injected-noise and per-element Philox paths and the record code compiled out and the evaluation cut into basic blocks
// (mlp_wide_body.h); instantiated for the MODE 2 shapes without a diagonal mass in mlp_wide_hmc_fast.hip.
template <int HT, int DT, int MODE, bool DIAGM, bool FAST = false>
__global__ __launch_bounds__(kBlock, 1) void mlp_wide_hmc_kernel(WideHmcArgs a) {
  constexpr bool EVAL_SCALED = false;  // (the chain kernel's THIN calls only: mlp_wide_body.h)
#include "mlp_wide_setup.inc"

  // The accepted position stays in a.x (in/out): read at the top of a transition, written back by the chains that
  // accept -- 16 DT registers fewer than carrying it, for one pass over the state per L + 1 evaluations.
  auto load_state = [&](float (&dst)[DT][16], int64_t row) {
#pragma unroll
    for (int td = 0; td < DT; ++td)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * td + 8 * q + 4 * h;
        if (quads && active && c0 + 3 < dim) {
          const float4 v = *reinterpret_cast<const float4*>(a.x + row * dim + c0);
          dst[td][4 * q] = v.x; dst[td][4 * q + 1] = v.y; dst[td][4 * q + 2] = v.z; dst[td][4 * q + 3] = v.w;
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) dst[td][4 * q + i] = (active && c0 + i < dim) ? a.x[row * dim + c0 + i] : 0.0f;
        }
      }
  };
  const bool has_mass = a.mass_kind != EBM_MASS_NONE;
  // diagonal mass: per coordinate from the (L2-resident) vector where it is used; 1 beyond dim
  auto mass_at = [&](int td, int r) -> float {
    const int c = 32 * td + row_of(r, h);
    return c < dim ? a.mass_diag[c] : 1.0f;
  };
  // 0.5 p^T M^-1 p over the whole chain (both K-halves), clamped to [0, 1e10]
  auto kinetic = [&](const float (&q)[DT][16]) -> float {
    float acc = 0.0f;
#pragma unroll
    for (int td = 0; td < DT; ++td)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float sq = q[td][r] * q[td][r];
        if constexpr (DIAGM) sq = sq / mass_at(td, r);
        acc += sq;
      }
    acc += __shfl_xor(acc, 32);
    float k = 0.5f * acc;
    if (!DIAGM && has_mass) k = k / a.mass_raw;
    return clamp_nanprop(k, 0.0f, 1e10f);
  };

  int until_keep = a.thin;
  int64_t keep_off = 0;
  float eps = a.eps;

  for (int tr = 0; tr < a.n_mh; ++tr) {
    if (a.eps_table) eps = a.eps_table[tr];
    const float half_eps = 0.5f * eps;
    int64_t smp = sample;  // nothing derived from the chain index is hoisted out of the transition loop and spilled
    asm volatile("" : "+v"(smp));

    // ---- momentum draw p ~ N(0, M)
    float p[DT][16];
#pragma unroll
    for (int td = 0; td < DT; ++td)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c0 = 32 * td + 8 * q + 4 * h;
        float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        if (c0 < dim) {
          if (!FAST && a.p_noise) {
            if (active)
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (c0 + i < dim) z[i] = a.p_noise[((int64_t)tr * a.n_chains + smp) * dim + c0 + i];
          } else if (quads) {  // the quad is exactly one Philox counter
            const F4 nrm = normal4_at(a.key, ((uint64_t)smp * (uint64_t)dim + (uint64_t)c0) >> 2, a.step0 + 2ull * (uint64_t)tr);
#pragma unroll
            for (int i = 0; i < 4; ++i) z[i] = nrm.v[i];
          } else if constexpr (FAST) {  // dim == 2: a chain's two elements are half a Philox counter
            if (td == 0 && q == 0) {
              const F4 nrm = normal4_at(a.key, (uint64_t)smp >> 1, a.step0 + 2ull * (uint64_t)tr);
              const bool odd = (smp & 1) != 0;
              z[0] = odd ? nrm.v[2] : nrm.v[0];
              z[1] = odd ? nrm.v[3] : nrm.v[1];
            }
          } else {
            uint64_t have = ~0ull;
            F4 nrm;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              const uint64_t e = (uint64_t)smp * (uint64_t)dim + (uint64_t)(c0 + i);
              if ((e >> 2) != have) {
                have = e >> 2;
                nrm = normal4_at(a.key, have, a.step0 + 2ull * (uint64_t)tr);
              }
              const int w = (int)(e & 3);
              z[i] = w == 0 ? nrm.v[0] : (w == 1 ? nrm.v[1] : (w == 2 ? nrm.v[2] : nrm.v[3]));
            }
          }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float v = (c0 + i < dim) ? z[i] : 0.0f;
          if constexpr (DIAGM) v *= sqrtf(mass_at(td, 4 * q + i));
          else if (has_mass) v *= a.mass_sqrt;
          p[td][4 * q + i] = v;
        }
      }
    // drift coefficient eps / max(m, 1e-10) (diagonal mass: formed once per transition)
    float em[DIAGM ? DT : 1][16];
    if constexpr (DIAGM) {
#pragma unroll
      for (int td = 0; td < DT; ++td)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float mr = mass_at(td, r);
          em[td][r] = eps / (mr < 1e-10f ? 1e-10f : mr);
        }
    }
    const float em_s = has_mass ? eps / a.mass_safe : eps;

    float xr[DT][16], f[DT][16];
    load_state(xr, smp);
#pragma unroll
    for (int td = 0; td < DT; ++td)
#pragma unroll
      for (int r = 0; r < 16; ++r) f[td][r] = 0.0f;
    float h0 = 0.0f, e_last = 0.0f, e_start = 0.0f;
    int done = 0, mode = 0;  // wave-uniform
    while (done <= a.n_leapfrog) {
      if (mode == 1) {  // first half kick + drift
#pragma unroll
        for (int td = 0; td < DT; ++td)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const float ph = __builtin_fmaf(half_eps, f[td][r], p[td][r]);
            p[td][r] = ph;
            float coef = em_s;
            if constexpr (DIAGM) coef = em[td][r];
            const float xn = __builtin_fmaf(coef, ph, xr[td][r]);
            xr[td][r] = (32 * td + row_of(r, h) < dim) ? xn : 0.0f;
          }
      }
      constexpr bool eval_energy_only = false, eval_block_cuts = false, eval_store_acts = false, eval_need_energy = true, eval_pin = false;
      [[maybe_unused]] float* const act_base = nullptr;
      [[maybe_unused]] constexpr uint32_t act_lane = 0;
      [[maybe_unused]] constexpr float act_seed = 1.0f;
      [[maybe_unused]] const auto eval_aux = [](auto) __attribute__((always_inline)) {};  // (nothing to hide in the tails' empty gaps)
      [[maybe_unused]] constexpr bool slab_more = true;  // MODE 3: every evaluation asks for the next one's first slab (drained at the end)
#include "mlp_wide_eval.inc"
      if (mode == 0) {  // H0 and the first (clamped) force
        h0 = clamp_nanprop(energy, -1e10f, 1e10f) + kinetic(p);
#pragma unroll
        for (int td = 0; td < DT; ++td)
#pragma unroll
          for (int r = 0; r < 16; ++r) f[td][r] = clamp_nanprop(-g[td][r], -1e6f, 1e6f);
        e_last = energy;
        e_start = energy;
        mode = 1;
        ++done;
      } else if (mode == 1) {
        // E finite => x finite and the gradient free of NaN (hmc_kernel.h); decided per WAVE because the literal path
        // re-runs the MFMA evaluation
        // (MODE 3: per WORKGROUP -- its evaluation has barriers inside, every wave must make the same number of them; for
        //  chains that are fine the literal path computes exactly what the fast path does)
        const bool all_fine = SLAB ? (__syncthreads_and(__builtin_fabsf(energy) < __builtin_inff() ? 1 : 0) != 0)
                                   : (bool)__all(__builtin_fabsf(energy) < __builtin_inff());
        if (all_fine) {
          float pz = 0.0f;
#pragma unroll
          for (int td = 0; td < DT; ++td)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float fn = __builtin_amdgcn_fmed3f(-g[td][r], -1e6f, 1e6f);
              const float pn = __builtin_fmaf(half_eps, fn, p[td][r]);
              f[td][r] = fn;
              p[td][r] = pn;
              pz = __builtin_fmaf(pn, 0.0f, pz);
            }
          if (pz != pz) {  // momentum overflow: x is finite, so f stands
#pragma unroll
            for (int td = 0; td < DT; ++td)
#pragma unroll
              for (int r = 0; r < 16; ++r) p[td][r] = nan_to_num0(p[td][r]);
          }
          e_last = energy;
          ++done;
        } else {  // literal semantics: NaN-propagating clamp, scrub, then re-evaluate on the scrubbed x
#pragma unroll
          for (int td = 0; td < DT; ++td)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const float fn = clamp_nanprop(-g[td][r], -1e6f, 1e6f);
              p[td][r] = nan_to_num0(__builtin_fmaf(half_eps, fn, p[td][r]));
              xr[td][r] = nan_to_num0(xr[td][r]);
            }
          mode = 2;
        }
      } else {  // mode 2: force and energy on the scrubbed position
#pragma unroll
        for (int td = 0; td < DT; ++td)
#pragma unroll
          for (int r = 0; r < 16; ++r) f[td][r] = clamp_nanprop(-g[td][r], -1e6f, 1e6f);
        e_last = energy;
        mode = 1;
        ++done;
      }
    }
    const float h1 = clamp_nanprop(e_last, -1e10f, 1e10f) + kinetic(p);

    // ---- Metropolis accept (samplers/hmc.py:277-292)
    const float dlt = clamp_nanprop(h0 - h1, -50.0f, 50.0f);
    float acc_p = expf(dlt);
    acc_p = (acc_p > 1.0f) ? 1.0f : acc_p;  // clamp_(max=1); NaN stays NaN and rejects
    float uu;
    if (!FAST && a.u) uu = active ? a.u[(int64_t)tr * a.n_chains + smp] : 2.0f;
    else uu = u01_half_open(pick(philox_at(a.key, (uint64_t)smp >> 2, a.step0 + 2ull * (uint64_t)tr + 1ull), (int)(smp & 3)));
    const bool accept = active && (uu < acc_p);
    if (accept) {
#pragma unroll
      for (int td = 0; td < DT; ++td)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = 32 * td + row_of(r, h);
          if (c < dim) a.x[smp * dim + c] = xr[td][r];
        }
    }
    const bool leader = active && h == 0;
    if (a.accept_mask && leader) a.accept_mask[(int64_t)tr * a.n_chains + smp] = accept ? 1 : 0;
    if (a.accept_count) {
      const unsigned long long b = __ballot(accept && leader);
      if (lane == 0 && b) atomicAdd(a.accept_count + tr, (uint32_t)__popcll(b));
    }
    if ((a.traj || (!FAST && a.diag_partials)) && --until_keep == 0) {
      until_keep = a.thin;
      if (active && !accept) load_state(xr, smp);  // a rejected chain records the position it stays at
      if (!FAST && a.diag_partials) {  // hmc.py:294-310: statistics of the state after the accept step, its (clamped) energy, the accept rate
        const int64_t wave_id = (int64_t)blockIdx.x * (kBlock / 64) + (threadIdx.x >> 6);
        const int kept = (int)(keep_off / dim);
        wave_record<DT>(a.diag_partials, a.diag_blocks, kept, wave_id, dim, xr, active, lane);
        wave_record_tail(a.diag_partials, a.diag_blocks, kept, wave_id, dim, clamp_nanprop(accept ? e_last : e_start, -1e10f, 1e10f),
                         active, accept, lane);
      }
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
    }
  }
  if constexpr (SLAB) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (the slab the last evaluation asked for)
}

template <int HT, int DT, bool DIAGM, bool FAST, int MODE_ = -1>
int launch_hmc_variant(const WideHmcArgs& a, hipStream_t st, const char* who) {
  constexpr int MODE = MODE_ >= 0 ? MODE_ : wide_mode(HT, DT);
  constexpr bool STREAM = MODE == 1;
  const size_t smem = wide_smem_bytes(HT, DT, MODE);
  if (STREAM && (reinterpret_cast<uintptr_t>(a.params) & 15) != 0)
    return fail(EBM_EINVAL, "%s: the MLP parameter block must be 16-byte aligned", who);
  static DeviceOnce attr_once;  // the LDS opt-in is a per-device function attribute
  if (attr_once.first()) {  // > 64 KiB of dynamic LDS needs the opt-in
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(mlp_wide_hmc_kernel<HT, DT, MODE, DIAGM, FAST>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  }
  const int64_t blocks = ceil_div64(a.n_chains, 32 * (kBlock / 64));
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "%s: too many chains for one launch", who);
  hipLaunchKernelGGL((mlp_wide_hmc_kernel<HT, DT, MODE, DIAGM, FAST>), dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch(who);
}

// the FAST instantiations live in mlp_wide_hmc_fast.hip
template <int HT, int DT>
int launch_hmc_fast(const WideHmcArgs& a, hipStream_t st, const char* who);
#define EBM_HMC_FAST_DECL(HTV, DTV) template <> int launch_hmc_fast<HTV, DTV>(const WideHmcArgs& a, hipStream_t st, const char* who);
EBM_HMC_FAST_DECL(2, 1) EBM_HMC_FAST_DECL(2, 2) EBM_HMC_FAST_DECL(2, 3) EBM_HMC_FAST_DECL(2, 4) EBM_HMC_FAST_DECL(4, 1) EBM_HMC_FAST_DECL(4, 2)
#undef EBM_HMC_FAST_DECL

// MODE 3 (mlp_wide_hmc_slab.hip): H = 128, dim 65 .. 128 with the W1 image at hand
template <int DT, bool DIAGM>
int launch_hmc_slab(const WideHmcArgs& a, hipStream_t st, const char* who);
#define EBM_HMC_SLAB_DECL(DTV, DM) template <> int launch_hmc_slab<DTV, DM>(const WideHmcArgs& a, hipStream_t st, const char* who);
EBM_HMC_SLAB_DECL(3, false) EBM_HMC_SLAB_DECL(3, true) EBM_HMC_SLAB_DECL(4, false) EBM_HMC_SLAB_DECL(4, true)
#undef EBM_HMC_SLAB_DECL

template <int HT, int DT, bool DIAGM>
int launch_hmc_one(const WideHmcArgs& a, hipStream_t st, const char* who) {
#ifndef EBM_MLP_F32LDS
  if constexpr (HT == 4 && DT >= 3) {
    if (a.w1_image && (reinterpret_cast<uintptr_t>(a.w1_image) & 15) == 0) return launch_hmc_slab<DT, DIAGM>(a, st, who);
  }
#endif
  if constexpr (wide_mode(HT, DT) == 2 && !DIAGM) {
    if (!a.p_noise && !a.u && !a.diag_partials && ((a.dim & 3) == 0 || a.dim == 2)) return launch_hmc_fast<HT, DT>(a, st, who);
  }
  return launch_hmc_variant<HT, DT, DIAGM, false>(a, st, who);
}

}  // namespace widemlp
}  // namespace ebm
