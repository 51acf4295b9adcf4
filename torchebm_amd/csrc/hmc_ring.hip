// HMC transition kernel for mixtures whose component means differ in ONE aligned block of four columns only (the `aux` mask of
// EBM_ENERGY_GMM a single bit; bit 0: a K-mode mixture of a plane embedded in a wider state -- BASELINE config 3's
// eight-mode ring), one lane per chain, dim 32, identity mass.  Round 4: its own kernel at FOUR waves per SIMD.
//
// Reference: torchebm/samplers/hmc.py:243-312 (transition loop, Metropolis accept),
// torchebm/integrators/leapfrog.py:156-185 (leapfrog, safe mode), core/base_integrator.py:875-889 (clamp / scrub).
//
// Why a kernel of its own.  The shared body of hmc_kernel.h keeps x, p, the force and the gradient of a 32-wide
// row in registers (128 + temporaries), carries both mixture bodies and every safe-mode path inline, and ran at
// 256 VGPRs = two waves per SIMD with 27 spilled registers; two waves issue plain VALU work at 75 % of the chip's
// rate (profiles/r02_valu_occupancy.txt).  This kernel is built around a 128-register budget:
//   * the state x and the momentum p are the only full rows in registers (64 VGPRs);
//   * ACTIVE columns are those of slot 0 on which the component means differ: all four, or -- the ring, whose means
//     sit in a plane -- columns 0 and 1 only (the kernel looks at the means itself: a wave-uniform branch between two
//     instantiations of the body).  Every other column is SHARED: all components agree on its mean, it drops out of
//     the responsibilities and its energy is that of one Gaussian;
//   * shared columns are held as y = x - mu_0 (exact when mu_0 is zero there, as on the ring; one rounding at load and
//     one at store otherwise -- the HMC state is a tolerance tier), so their force -y / sigma^2 is never formed: the
//     kick is ONE packed FMA, p += (-kick / sigma^2) y, and a leapfrog step costs a shared pair of columns two packed
//     FMAs -- the floor for a leapfrog step.  Only the active forces are carried from step to step (and from
//     transition to transition: an accepted proposal's end-of-trajectory force starts the next trajectory);
//   * the accepted state is parked in a lane-private LDS slot during the proposal (32 KiB per workgroup, four
//     workgroups per CU), the carried active force next to it;
//   * every constant of the step loop is a scalar register: the active means pre-scaled by log2(e) / sigma^2 (the
//     logits come out in base 2: v_exp_f32 directly), in the two pairings the two passes use, and the logit offsets;
//   * safe mode costs running NaN-propagating maxima (v_maximum3_f32, one per packed pair).  The reference clamps
//     the force to +-1e6 and scrubs non-finite x / p after every step; both are the identity while every |f| <= 1e6
//     and p starts below 1e30 (the force is clamped, so p cannot leave the finite range within a trajectory), and a
//     non-finite coordinate or logit makes a force NaN / inf.  A chain whose trajectory ends outside those bounds is
//     REDONE from the parked state by the literal sequence (NaN-propagating clamps, half kicks, scrub after every
//     step, force re-evaluated on the scrubbed state) -- cold code behind a wave-level branch.
#include "hmc_lane.h"

namespace ebm {
namespace hmc {

using namespace lane;

namespace {

constexpr int kTabFloats = 32 + 8 + 32;  // raw slot-0 means [8][4], log-weights [8], row 0 of the means [32]
constexpr int kLdsHead = kTabFloats + 8; // ... and the eight logit offsets behind the table

// ACT: active columns (2 or 4), pairs 0 .. ACT/2 - 1 of the row.
// DIAG: per-block diagnostics records at the kept transitions (diag.h), a compile-time switch: the call into diag::emit
// costs registers around it.
// slot: the 4-column block the means differ in.  The row is held with that block FIRST: register block i holds column block
// blk(i) (0 <-> slot swapped) -- every per-column operation is symmetric in the columns, so only the addresses know
// (loads, stores, the Philox counters of the momentum draw, the staged tables).
template <int ACT, bool DIAG>
__device__ __forceinline__ void slot1_body(const HmcArgs& a, float* const tab, const int slot) {
  constexpr int AP = ACT / 2;  // active pairs
  const auto blk = [slot](int i) { return i == 0 ? slot : (i == slot ? 0 : i); };
  // The lane's chain index is the ONLY per-lane address register that lives through the kernel: every global address
  // is formed from it where it is used (chain_now() hides it from the optimiser, which otherwise hoists row offsets,
  // Philox counters and pointers out of the transition loop -- a dozen 64-bit registers, spilled and reloaded).
  const uint32_t chain32 = blockIdx.x * (uint32_t)kBlock + threadIdx.x;  // n_chains < 2^32 (checked by the launcher, hmc.hip)
  const bool active = (int64_t)chain32 < a.n_chains;
  auto chain_now = [&]() -> uint64_t {
    uint32_t c = chain32;
    asm volatile("" : "+v"(c));
    return active ? (uint64_t)c : 0ull;
  };
  const int park0 = kLdsHead + 4 * (int)threadIdx.x;
  const int fpark = kLdsHead + NV * 4 * kBlock + 4 * (int)threadIdx.x;

  // ---- constants of the step loop (wave-uniform: scalar registers)
  const float invs2 = a.energy.s1, inv2s2 = a.energy.s0;
  float m2[8][ACT];  // mu_k[i] * log2(e) / sigma^2
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int i = 0; i < ACT; ++i) m2[k][i] = to_sgpr(tab[4 * k + i] * (invs2 * kLog2e));
  // (logw_k - |mu_k[active]|^2 / (2 sigma^2)) * log2(e), the x-independent part of the base-2 logit: in the LDS table,
  // read into vector-register pairs at the top of every trajectory (a broadcast read; no register lives across the
  // momentum draw for them)
  if (threadIdx.x < 8) {
    const int k = threadIdx.x;
    float nrm = 0.0f;
#pragma unroll
    for (int i = 0; i < ACT; ++i) nrm = __builtin_fmaf(tab[4 * k + i], tab[4 * k + i], nrm);
    tab[kTabFloats + k] = __builtin_fmaf(-nrm, inv2s2, tab[32 + k]) * kLog2e;
  }
  __syncthreads();

  // active force -dE/dx at the active columns:  f = (sum_k r_k mu_k - x) / sigma^2,  r = softmax of the logits
  //   l_k = c2_k + sum_i x_i m2_ki   (components in packed pairs (2q, 2q+1); x_i broadcast by op_sel)
  auto active_force = [&](const v2f (&XA)[AP], const v2f (&C2)[4], v2f (&FA)[AP]) {
    v2f lp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v2f t = C2[q];
#pragma unroll
      for (int i = 0; i < ACT; ++i) {
        const float xi = (i & 1) ? XA[i >> 1].y : XA[i >> 1].x;
        t = pk_fma(splat(xi), v2f{m2[2 * q][i], m2[2 * q + 1][i]}, t);
      }
      lp[q] = t;
    }
    float top = __builtin_fmaxf(__builtin_fmaxf(lp[0].x, lp[0].y), lp[1].x);  // a NaN logit resurfaces in the sum
    top = __builtin_fmaxf(__builtin_fmaxf(top, lp[1].y), lp[2].x);
    top = __builtin_fmaxf(__builtin_fmaxf(top, lp[2].y), lp[3].x);
    top = __builtin_fmaxf(top, lp[3].y);
    v2f w[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const v2f t = lp[q] - splat(top);
      w[q] = v2f{__builtin_amdgcn_exp2f(t.x), __builtin_amdgcn_exp2f(t.y)};
    }
    const v2f s2 = (w[0] + w[1]) + (w[2] + w[3]);
    const float sum = s2.x + s2.y;  // in [1, 8] for finite logits
    // 1 / (sigma^2 log2 e) is folded into m2: sum_k w_k m2_k = (log2 e / sigma^2) sum_k w_k mu_k
    const float s = __builtin_amdgcn_rcpf(sum) * kLn2;
    // weighted means, columns in packed pairs (i, i+1), w_k broadcast by op_sel
#pragma unroll
    for (int h = 0; h < AP; ++h) {
      v2f acc = splat(w[0].x) * v2f{m2[0][2 * h], m2[0][2 * h + 1]};
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        const float wk = (k & 1) ? w[k >> 1].y : w[k >> 1].x;
        acc = pk_fma(splat(wk), v2f{m2[k][2 * h], m2[k][2 * h + 1]}, acc);
      }
      FA[h] = pk_fma(acc, splat(s), XA[h] * splat(-invs2));
    }
  };

  // E(x) in the reference's difference form (the form the shared body evaluates H0 / H1 in):
  //   sum_{shared d} (x_d - mu_0d)^2 / (2 sigma^2) - logsumexp_k(logw_k - |x_act - mu_k,act|^2 / (2 sigma^2))
  auto energy_exact = [&](const v2f (&X)[NP]) -> float {
    float logit[8];
    // Two table rows in flight at a time: left alone the scheduler issues all ten LDS reads first (40 registers).  The
    // read offset of the next pair is tied to the logits of this one (an empty asm: no instruction).
    int toff = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 m = *reinterpret_cast<const float4*>(&tab[4 * k + toff]);
      const v2f da = X[0] - v2f{m.x, m.y};
      v2f d2 = da * da;
      if constexpr (ACT == 4) {
        const v2f db = X[1] - v2f{m.z, m.w};
        d2 = pk_fma(db, db, d2);
      }
      logit[k] = __builtin_fmaf(-(d2.x + d2.y), inv2s2, tab[32 + k + toff]);
      if (k & 1) asm volatile("" : "+v"(toff) : "v"(logit[k]), "v"(logit[k - 1]));
    }
    float top = logit[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) top = __builtin_fmaxf(top, logit[k]);
    // torch.logsumexp shifts by the maximum unless that is infinite (then by 0): every distance overflowing gives
    // logsumexp = -inf, E = +inf -- not NaN (the energy the reference clamps to 1e10 in H)
    top = __builtin_fabsf(top) == __builtin_inff() ? 0.0f : top;
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += __expf(logit[k] - top);
    v2f sq = {0.0f, 0.0f}, sq_b = {0.0f, 0.0f};
#pragma unroll
    for (int j = AP; j < NP; ++j) {  // the shared columns are held as x - mu_0
      if ((j - AP) & 1) sq_b = pk_fma(X[j], X[j], sq_b);
      else sq = pk_fma(X[j], X[j], sq);
    }
    sq += sq_b;
    return __builtin_fmaf(sq.x + sq.y, inv2s2, -(top + logf(sum)));
  };

  // K(p) = 0.5 |p|^2 clamped to [0, 1e10]  (samplers/hmc.py:136-159, :251-254)
  auto kinetic = [&](const v2f (&P)[NP]) -> float {
    v2f acc = P[0] * P[0], acc_b = P[1] * P[1];
#pragma unroll
    for (int j = 2; j < NP; j += 2) {
      acc = pk_fma(P[j], P[j], acc);
      acc_b = pk_fma(P[j + 1], P[j + 1], acc_b);
    }
    acc += acc_b;
    return clamp_nanprop(0.5f * (acc.x + acc.y), 0.0f, 1e10f);
  };

  // full rows, 16-byte aligned: whole float4 accesses; lanes past the last chain hold zeros and never store
  auto load_row = [&](const float* __restrict__ src, v2f (&R)[NP]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (active) q = *reinterpret_cast<const float4*>(src + 4 * blk(v));
      R[2 * v] = v2f{q.x, q.y};
      R[2 * v + 1] = v2f{q.z, q.w};
    }
  };
  // shared columns <-> y = x - mu_0 (row 0 of the means, from the LDS table)
  auto shift = [&](v2f (&R)[NP], float sign) {
#pragma unroll
    for (int j = AP; j < NP; ++j) {
      const float2 mu = *reinterpret_cast<const float2*>(&tab[40 + 2 * j]);
      R[j] = pk_fma(splat(sign), v2f{mu.x, mu.y}, R[j]);  // sign = +-1: the product is exact
    }
  };
  auto store_row = [&](float* __restrict__ dst, const v2f (&R)[NP]) {
    v2f O[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) O[j] = R[j];
    shift(O, 1.0f);
    if (active) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
        *reinterpret_cast<float4*>(dst + 4 * blk(v)) = make_float4(O[2 * v].x, O[2 * v].y, O[2 * v + 1].x, O[2 * v + 1].y);
    }
  };
  auto unpark = [&](v2f (&R)[NP]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float4 q = *reinterpret_cast<const float4*>(&hmc_smem[park0 + v * (4 * kBlock)]);
      R[2 * v] = v2f{q.x, q.y};
      R[2 * v + 1] = v2f{q.z, q.w};
    }
  };
  v2f X[NP];
  load_row(a.x + chain_now() * D, X);
  shift(X, -1.0f);
  auto draw_momentum = [&](int t, v2f (&P)[NP]) {
    if (a.p_noise) {
      load_row(a.p_noise + ((uint64_t)t * (uint64_t)a.n_chains + chain_now()) * D, P);
    } else {
      const uint64_t step = a.step0 + 2ull * (uint64_t)t;
      // (the chain index hidden from the optimiser: the counter words and the first Philox multiply of all eight calls
      //  are invariant across transitions -- hoisted out of the loop they are 16 registers, spilled and reloaded)
      uint32_t c = chain32;
      asm volatile("" : "+v"(c));
      const uint64_t g0 = (uint64_t)c * (uint64_t)(D / 4);  // Philox counter of the row's first float4
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const F4 n = normal4_at(a.key, g0 + (uint64_t)blk(v), step);
        P[2 * v] = v2f{n.v[0], n.v[1]};
        P[2 * v + 1] = v2f{n.v[2], n.v[3]};
        // two counters at a time: eight interleaved Philox chains cost more registers than the budget has
        if (v & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  int until_keep = a.thin;
  int keep = 0;
  float eps = a.eps;
  float e_cur = 0.0f, e_keep = 0.0f;
  v2f FA[AP];  // active force at the state the chain holds
#pragma unroll
  for (int h = 0; h < AP; ++h) FA[h] = v2f{0.0f, 0.0f};

  // t = -1 is a pseudo-transition (zero momentum, zero step size, one leapfrog step, always "accepted", nothing
  // written): x + 0 * p is x bit for bit, so it leaves the energy and the carried force of the initial state.
  for (int t = -1; t < a.n_mh; ++t) {
    const bool init = t < 0;
    if (a.eps_table && !init) eps = to_sgpr(a.eps_table[t]);  // wave-uniform: a scalar register
    const float eps_t = init ? 0.0f : eps;
    const float half_eps = 0.5f * eps_t;
    const int n_lf = init ? 1 : a.n_leapfrog;

    v2f P[NP];
    if (init) {
#pragma unroll
      for (int j = 0; j < NP; ++j) P[j] = v2f{0.0f, 0.0f};
    } else {
      draw_momentum(t, P);
    }
    float uu;
    if (init) uu = -1.0f;
    else if (a.u) uu = active ? a.u[(uint64_t)t * (uint64_t)a.n_chains + chain_now()] : 2.0f;
    else {
      uint32_t c = chain32;
      asm volatile("" : "+v"(c));
      uu = u01_half_open(pick(philox_at(a.key, (uint64_t)(c >> 2), a.step0 + 2ull * (uint64_t)t + 1ull), (int)(c & 3)));
    }
    const float h0 = clamp_nanprop(e_cur, -1e10f, 1e10f) + kinetic(P);

    // park the accepted state and its active force
#pragma unroll
    for (int v = 0; v < NV; ++v)
      *reinterpret_cast<float4*>(&hmc_smem[park0 + v * (4 * kBlock)]) = make_float4(X[2 * v].x, X[2 * v].y, X[2 * v + 1].x, X[2 * v + 1].y);
    // ... and the scalars that are only needed again behind the trajectory: the chain's active force and energy,
    // H0 and the accept uniform (four active columns: the force is re-evaluated on a rejection instead -- the slot
    // holds four floats, and a CU's LDS holds four workgroups only at nine float4 per lane)
    if constexpr (AP == 1) *reinterpret_cast<float4*>(&hmc_smem[fpark]) = make_float4(FA[0].x, FA[0].y, h0, uu);
    else *reinterpret_cast<float4*>(&hmc_smem[fpark]) = make_float4(e_cur, 0.0f, h0, uu);

    // the logit offsets as vector-register pairs (the packed FMA that adds them already takes the means from scalar
    // registers)
    v2f C2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float2 c = *reinterpret_cast<const float2*>(&tab[kTabFloats + 2 * q]);
      C2[q] = v2f{c.x, c.y};
    }

    // ---- the trajectory, common path.  Running NaN-propagating maxima: m_p of the entry momentum, m_y of the shared
    //      coordinates a force is taken at, m_f of the active forces
    float m_p = 0.0f, m_y = 0.0f, m_f = 0.0f;
#pragma unroll
    for (int j = 0; j < NP; ++j) m_p = max3np(m_p, __builtin_fabsf(P[j].x), __builtin_fabsf(P[j].y));
    asm volatile("" : "+v"(m_p));  // here, not sunk to its use behind the trajectory (see h0)
    {
      const v2f H2 = splat(half_eps), HN = splat(half_eps * -invs2);
#pragma unroll
      for (int h = 0; h < AP; ++h) {
        m_f = max3np(m_f, __builtin_fabsf(FA[h].x), __builtin_fabsf(FA[h].y));
        P[h] = pk_fma(H2, FA[h], P[h]);
      }
#pragma unroll
      for (int j = AP; j < NP; ++j) {
        m_y = max3np(m_y, __builtin_fabsf(X[j].x), __builtin_fabsf(X[j].y));
        P[j] = pk_fma(HN, X[j], P[j]);
      }
    }
    const v2f E2 = splat(eps_t);
    for (int l = 0; l < n_lf; ++l) {
      const float kick = l + 1 >= n_lf ? half_eps : eps_t;  // the next step's first half kick rides along
      const v2f K2 = splat(kick), KN = splat(kick * -invs2);
#pragma unroll
      for (int j = 0; j < NP; ++j) X[j] = pk_fma(E2, P[j], X[j]);
      v2f XA[AP];
#pragma unroll
      for (int h = 0; h < AP; ++h) XA[h] = X[h];
      active_force(XA, C2, FA);
#pragma unroll
      for (int h = 0; h < AP; ++h) {
        m_f = max3np(m_f, __builtin_fabsf(FA[h].x), __builtin_fabsf(FA[h].y));
        P[h] = pk_fma(K2, FA[h], P[h]);
      }
#pragma unroll
      for (int j = AP; j < NP; ++j) {
        m_y = max3np(m_y, __builtin_fabsf(X[j].x), __builtin_fabsf(X[j].y));
        P[j] = pk_fma(KN, X[j], P[j]);
      }
    }
    float e1 = energy_exact(X);
    // Anything the reference's safe mode would have touched?  (a clamped force, a non-finite coordinate / logit /
    // momentum, a non-finite energy at the end: leapfrog.py:165-185 clamps and scrubs.)
    const bool bad = !init && (!(m_y * invs2 <= 1e6f) || !(m_f <= 1e6f) || !(m_p < 1e30f) || !(__builtin_fabsf(e1) < __builtin_inff()));
    if (__builtin_expect(bad, 0)) {
      // ---- the literal sequence from the parked state:  per step  f = clamp(-dE/dx(x)); p += eps/2 f; x += eps p;
      //      f' = clamp(-dE/dx(x)); p += eps/2 f'; scrub x, p
      unpark(X);
      draw_momentum(t, P);
      // -dE/dx as autograd returns it for E = -logsumexp_k(logw_k - |x - mu_k|^2 / (2 sigma^2)): when every squared
      // distance overflows all logits are -inf and the softmax -- the whole gradient row -- is NaN.  E(x) = +inf (or NaN)
      // says so; the active-column form above would not notice an overflow in the shared columns.
      auto literal_force = [&](v2f (&G)[AP]) -> bool {
        v2f XA[AP];
#pragma unroll
        for (int h = 0; h < AP; ++h) XA[h] = X[h];
        active_force(XA, C2, G);
        return !(energy_exact(X) < __builtin_inff());
      };
      auto half_kick = [&]() {
        v2f G[AP];
        const bool blown = literal_force(G);
        const float nan = __builtin_nanf("");
#pragma unroll
        for (int h = 0; h < AP; ++h) {
          P[h].x = __builtin_fmaf(half_eps, blown ? nan : clamp_nanprop(G[h].x, -1e6f, 1e6f), P[h].x);
          P[h].y = __builtin_fmaf(half_eps, blown ? nan : clamp_nanprop(G[h].y, -1e6f, 1e6f), P[h].y);
        }
#pragma unroll
        for (int j = AP; j < NP; ++j) {
          const v2f F = X[j] * splat(-invs2);
          P[j].x = __builtin_fmaf(half_eps, blown ? nan : clamp_nanprop(F.x, -1e6f, 1e6f), P[j].x);
          P[j].y = __builtin_fmaf(half_eps, blown ? nan : clamp_nanprop(F.y, -1e6f, 1e6f), P[j].y);
        }
      };
      for (int l = 0; l < n_lf; ++l) {
        for (int h = 0; h < 2; ++h) {  // (a loop: ONE inlined copy of the kick)
          half_kick();
          if (h == 0) {
#pragma unroll
            for (int j = 0; j < NP; ++j) X[j] = pk_fma(E2, P[j], X[j]);
          }
        }
        // nan_to_num_ on x and p; the shared columns are scrubbed as x = y + mu_0 (a NaN becomes x = 0, not x = mu_0)
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          v2f xs = X[j], mu = v2f{0.0f, 0.0f};
          if (j >= AP) {
            const float2 m0 = *reinterpret_cast<const float2*>(&tab[40 + 2 * j]);
            mu = v2f{m0.x, m0.y};
            xs = xs + mu;
          }
          xs = v2f{nan_to_num0(xs.x), nan_to_num0(xs.y)};
          X[j] = (j >= AP) ? xs - mu : xs;
          P[j] = v2f{nan_to_num0(P[j].x), nan_to_num0(P[j].y)};
        }
      }
      // the force the next trajectory starts from (the reference re-evaluates it on the scrubbed state)
      const bool blown = literal_force(FA);
#pragma unroll
      for (int h = 0; h < AP; ++h)
        FA[h] = blown ? splat(__builtin_nanf("")) : v2f{clamp_nanprop(FA[h].x, -1e6f, 1e6f), clamp_nanprop(FA[h].y, -1e6f, 1e6f)};
      e1 = energy_exact(X);
    }
    const float h1 = clamp_nanprop(e1, -1e10f, 1e10f) + kinetic(P);
    const float4 parked = *reinterpret_cast<const float4*>(&hmc_smem[fpark]);
    const float h0_ = parked.z, uu_ = parked.w;
    if constexpr (AP == 1) e_keep = e_cur; else e_keep = parked.x;

    // ---- Metropolis accept (samplers/hmc.py:277-292)
    const float dlt = clamp_nanprop(h0_ - h1, -50.0f, 50.0f);
    float acc_p = expf(dlt);
    acc_p = (acc_p > 1.0f) ? 1.0f : acc_p;  // clamp_(max=1); NaN stays NaN and rejects
    const bool accept = init || (active && (uu_ < acc_p));
    if (accept) {
      e_cur = e1;
    } else {  // rejected: bring the parked state and its force back
      e_cur = e_keep;
      unpark(X);
      if constexpr (AP == 1) {
        FA[0] = v2f{parked.x, parked.y};
      } else {
        v2f XA[AP];
#pragma unroll
        for (int h = 0; h < AP; ++h) XA[h] = X[h];
        active_force(XA, C2, FA);  // finite state of a finished trajectory's start: inside the clamp
      }
    }
    if (init) continue;

    if (a.accept_mask && active) a.accept_mask[(uint64_t)t * (uint64_t)a.n_chains + chain_now()] = accept ? 1 : 0;
    if (a.accept_count) {  // wavefront-level count, one atomic per wave
      const unsigned long long b = __ballot(accept && active);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(a.accept_count + t, (uint32_t)__popcll(b));
    }
    if ((a.traj != nullptr || DIAG) && --until_keep == 0) {
      until_keep = a.thin;
      if (a.traj) store_row(a.traj + (chain_now() * (uint64_t)a.n_kept + (uint64_t)keep) * D, X);
      if constexpr (DIAG) {
        // samplers/hmc.py:294-310: population mean / var, mean of the clamped energy of the state the chain holds now,
        // acceptance rate.  The tile is the parking area of the state -- dead until the next transition parks again and
        // exactly one block of rows wide; the barrier: a slower wave may still have to bring its parked state back.
        float* const tile = hmc_smem + kLdsHead;
        float* const scratch = tile + (NV + 1) * 4 * kBlock;
        __syncthreads();
        {
          v2f O[NP];
#pragma unroll
          for (int j = 0; j < NP; ++j) O[j] = X[j];
          shift(O, 1.0f);
#pragma unroll
          for (int v = 0; v < NV; ++v)
            *reinterpret_cast<float4*>(tile + (int)threadIdx.x * D + 4 * blk(v)) = make_float4(O[2 * v].x, O[2 * v].y, O[2 * v + 1].x, O[2 * v + 1].y);
        }
        const int64_t left = a.n_chains - (int64_t)blockIdx.x * kBlock;
        const int valid = (left >= kBlock ? kBlock : (left > 0 ? (int)left : 0)) * D;
        diag::emit(a.diag, keep, tile, scratch, valid, D, active ? clamp_nanprop(e_cur, -1e10f, 1e10f) : 0.0f,
                   (accept && active) ? 1.0f : 0.0f);
      }
      ++keep;
    }
  }
  store_row(a.x + chain_now() * D, X);
}

}  // namespace

// ACT = 2: the means differ in columns 0 and 1 only (the ring); ACT = 4: anywhere in columns 0..3.  Both instantiations
// are launched, each returns at once unless the mixture is its own (one kernel per body: each gets the whole
// register budget -- with both bodies in one kernel the allocator spilled).
template <int ACT, bool DIAG>
__global__ __launch_bounds__(kBlock, 4) void hmc_slot1_kernel(HmcArgs a) {
  const int slot = gmm_single_slot(a.energy);
  if (slot < 0) return;  // no or several slots: the dense kernel, launched behind this one, does the work
  // ---- LDS: [table | parked state, [v][thread] float4 | parked active force, [thread] float4]
  float* const tab = hmc_smem;
  const int K = a.energy.n_comp;
  for (int i = threadIdx.x; i < kTabFloats; i += kBlock) {
    float v;
    if (i < 32) {
      const int k = i >> 2, kk = k < K ? k : K - 1;  // padding components repeat the last row, their log-weight is -inf
      v = a.energy.dev0[kk * D + 4 * slot + (i & 3)];
    } else if (i < 40) {
      v = (i - 32) < K ? a.energy.dev1[i - 32] : -__builtin_inff();
    } else {  // row 0 of the means in REGISTER order: position p holds column 4 blk(p / 4) + p % 4
      const int pos = i - 40, b = pos >> 2;
      v = a.energy.dev0[4 * (b == 0 ? slot : (b == slot ? 0 : b)) + (pos & 3)];
    }
    tab[i] = v;
  }
  __syncthreads();
  // columns 2 and 3 shared by all components (the ring: a mixture of a plane)?  wave-uniform
  bool plane = true;
  for (int k = 1; k < 8; ++k) plane = plane && tab[4 * k + 2] == tab[2] && tab[4 * k + 3] == tab[3];
  if ((__builtin_amdgcn_readfirstlane((int)plane) != 0) != (ACT == 2)) return;
  slot1_body<ACT, DIAG>(a, tab, slot);
}

// Launched IN FRONT of the dense kernel when the energy carries an active-column mask: the kernel whose body does not
// match the mask returns at once (a wave-uniform read of the mask: no host read of device memory).
bool hmc_slot1_applies(const ebm_energy_t& e, const rows::Geometry& geo, int32_t mass_kind) {
  return e.kind == EBM_ENERGY_GMM && e.aux != nullptr && e.n_comp >= 1 && e.n_comp <= 8 && geo.G == 1 && geo.NV == 8 &&
         geo.full && mass_kind == EBM_MASS_NONE;  // (one lane per chain; the launcher refuses 2^32 chains and more)
}

// a.diag.partials != nullptr: the records of the lane-group layout (diag::plan over kBlock rows per workgroup)
void launch_slot1(dim3 grid, hipStream_t st, HmcArgs a) {
  size_t smem = ((size_t)kLdsHead + (size_t)(NV + 1) * 4 * kBlock) * sizeof(float);
  if (a.diag.partials) {
    smem += (size_t)diag::scratch_floats(a.diag.S) * sizeof(float);
    hipLaunchKernelGGL((hmc_slot1_kernel<2, true>), grid, dim3(kBlock), smem, st, a);
    hipLaunchKernelGGL((hmc_slot1_kernel<4, true>), grid, dim3(kBlock), smem, st, a);
  } else {
    hipLaunchKernelGGL((hmc_slot1_kernel<2, false>), grid, dim3(kBlock), smem, st, a);
    hipLaunchKernelGGL((hmc_slot1_kernel<4, false>), grid, dim3(kBlock), smem, st, a);
  }
}

}  // namespace hmc
}  // namespace ebm
