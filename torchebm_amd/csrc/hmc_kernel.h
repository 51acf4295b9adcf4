// HMC transition kernel template (included by the per-energy translation units hmc_*.hip, which
// only exist so that the energies compile in parallel).  See hmc.hip for the entry point.
#pragma once
#include "rows.h"

namespace ebm {
namespace hmc {
using namespace rows;

struct HmcArgs {
  float* x;
  int64_t n_chains;
  int32_t dim;
  int32_t n_mh;
  int32_t n_leapfrog;
  float eps;
  const float* eps_table;
  int32_t mass_kind;
  float mass_raw, mass_sqrt, mass_safe;  // scalar mass forms
  const float* mass_diag;
  int32_t thin;
  int32_t n_kept;
  float* traj;
  uint8_t* accept_mask;
  uint32_t* accept_count;
  const float* p_noise;
  const float* u;
  RngKey key;
  uint64_t step0;
  EnergyParams energy;
  int param_floats;
  int park_offset_floats;  // start of the lane-private parking slots in dynamic LDS
  diag::DiagArgs diag;     // per-block diagnostics records at the kept transitions (null: off)
  int diag_offset_floats;  // start of the diagnostics tile in dynamic LDS
};

extern __shared__ __attribute__((aligned(16))) float hmc_smem[];

// L leapfrog steps in safe mode.  On entry f = clamp(-dE/dx) at x; on exit x, p are the
// proposal, f the clamped force there, and the return value is E(x).
//  * The clamped force at the end of a step is bit-identical to the one the reference
//    recomputes at the start of the next step, so it is carried over.
//  * torch's nan_to_num_ is the identity on finite values and clamp_ only differs from a plain
//    median-of-three on NaN, so the common path never touches either: every energy here has the
//    property "E(x) finite  =>  every x_i and every dE/dx_i is free of NaN, x is finite" (each
//    coordinate enters E through sums and products only), which turns 2 x 4NV finiteness tests and
//    4NV NaN-propagating clamps into ONE compare on the group-reduced energy plus v_med3 clamps.
//    Momentum can still overflow on its own; sum(p_i * 0) is NaN exactly when some p_i is not
//    finite (packed FMAs).  Lane groups that fail either test take the literal path: NaN-propagating
//    clamp, scrub, and the force re-evaluation the reference then performs on the scrubbed x.
//  * Energies with HAS_GRAD_ONLY (mixture) skip the energy on all but the last step; their
//    grad_only() returns a value with the same "finite => clean" property.
//  * Merged kicks.  The second half kick of step l and the first half kick of step l + 1 use the SAME clamped
//    force, so between two evaluations the momentum moves by ONE fused multiply-add with the whole step size
//    (p + eps f instead of (p + eps/2 f) + eps/2 f: one rounding less, in the tolerance tier like every FMA of
//    this kernel); the trajectory starts and ends with a half kick.  4 NV operations per step and -- the force is
//    consumed as soon as it exists -- 4 NV registers less across the evaluation.  If the merged kick leaves the
//    finite range the step is redone literally (half kick, scrub, half kick).
template <bool HAS_MASS, class En, class LaneT>
__device__ __forceinline__ float leapfrog_steps(const En& en, const LaneT& L, Slice<LaneT::NV>& x,
                                                Slice<LaneT::NV>& p, Slice<LaneT::NV>& f,
                                                const Slice<LaneT::NV>& m_safe, float eps, float half_eps,
                                                int n_steps, float e_in, bool init = false) {
  constexpr int NV = LaneT::NV;
  typedef float v2f __attribute__((ext_vector_type(2)));
  float e = e_in;
  // first half kick (fused multiply-adds: one rounding where the reference's eager ops take two -- HMC states are
  // a tolerance tier anyway: energy and gradient sums run in another order than torch's)
  bool tame = true;  // every |p_i| < 1e30 on entry
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      p.a[v][i] = __builtin_fmaf(half_eps, f.a[v][i], p.a[v][i]);
      tame = tame && (__builtin_fabsf(p.a[v][i]) < 1e30f);  // false for NaN / inf too
    }
  // Momentum moves by at most eps * 1e6 per step (the force is clamped), so a momentum that starts below 1e30
  // cannot leave the finite range within any trajectory a float step count can express: the per-step finiteness
  // test of the momentum (the reference's nan_to_num_ on p) is only kept for lane groups that start outside.
  const bool watch_p = group_any<LaneT::G>(!tame);
  for (int l = 0; l < n_steps; ++l) {
    const bool last = l + 1 >= n_steps;
    const float kick = last ? half_eps : eps;  // the next step's first half kick rides along
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float xn;
        if constexpr (HAS_MASS) xn = __builtin_fmaf(m_safe.a[v][i], p.a[v][i], x.a[v][i]);  // m_safe holds eps / max(m, 1e-10) here
        else xn = __builtin_fmaf(eps, p.a[v][i], x.a[v][i]);
        x.a[v][i] = L.ok(v, i) ? xn : 0.0f;
      }
    Slice<NV> g;
    float chk;  // group-uniform; finite => x finite, g free of NaN
    bool have_e = true;
    if constexpr (En::HAS_GRAD_ONLY) {
      if (!last && en.grad_only_ready()) {
        chk = en.grad_only(L, x, g);
        have_e = false;
      } else {
        chk = e = en.template eval<true>(L, x, g);
      }
    } else {
      chk = e = en.template eval<true>(L, x, g);
    }
    bool literal = !(__builtin_fabsf(chk) < __builtin_inff());
    if (!literal) {
      if (!watch_p) {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const float fc = __builtin_amdgcn_fmed3f(-g.a[v][i], -1e6f, 1e6f);
            const float pc = __builtin_fmaf(kick, fc, p.a[v][i]);
            f.a[v][i] = fc;
            p.a[v][i] = L.ok(v, i) ? pc : 0.0f;
          }
      } else {
        v2f pz = {0.0f, 0.0f};
        Slice<NV> pn;
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int i = 0; i < 4; i += 2) {
            const float f0 = __builtin_amdgcn_fmed3f(-g.a[v][i], -1e6f, 1e6f);
            const float f1 = __builtin_amdgcn_fmed3f(-g.a[v][i + 1], -1e6f, 1e6f);
            const float p0 = __builtin_fmaf(kick, f0, p.a[v][i]);
            const float p1 = __builtin_fmaf(kick, f1, p.a[v][i + 1]);
            f.a[v][i] = f0;
            f.a[v][i + 1] = f1;
            pn.a[v][i] = L.ok(v, i) ? p0 : 0.0f;
            pn.a[v][i + 1] = L.ok(v, i + 1) ? p1 : 0.0f;
            pz = __builtin_elementwise_fma(v2f{p0, p1}, v2f{0.0f, 0.0f}, pz);
          }
        const float pchk = pz.x + pz.y;
        if (group_any<LaneT::G>(pchk != pchk)) {  // momentum left the finite range (x is finite, so f stands):
#pragma unroll                                    // the literal sequence -- half kick, scrub, the next step's half kick
          for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
              float q = nan_to_num0(L.ok(v, i) ? __builtin_fmaf(half_eps, f.a[v][i], p.a[v][i]) : 0.0f);
              if (!last) q = __builtin_fmaf(half_eps, f.a[v][i], q);
              p.a[v][i] = q;
            }
        } else {
          p = pn;
        }
      }
    } else if (init) {
      // The pseudo-transition on a state that is not finite (or whose energy is not): NOTHING is scrubbed -- the reference
      // evaluates model(x) at the top of the first transition on the state as it was handed over (samplers/hmc.py:243-256), so
      // the carried energy is the raw one (NaN rejects every proposal, an infinite one is clamped in H0) and the chain keeps
      // its state until a proposal is accepted; the force takes the NaN-propagating clamp of the first half kick.
      if (!have_e) e = en.template eval<true>(L, x, g);
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) f.a[v][i] = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);
    } else {  // rare: literal semantics
      if (!have_e) e = en.template eval<true>(L, x, g);
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float fn = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);
          const float pl = __builtin_fmaf(half_eps, fn, p.a[v][i]);
          p.a[v][i] = nan_to_num0(L.ok(v, i) ? pl : 0.0f);
          x.a[v][i] = nan_to_num0(x.a[v][i]);
        }
      e = en.template eval<true>(L, x, g);  // the force the next step starts from, on the scrubbed x
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          f.a[v][i] = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);
          if (!last) p.a[v][i] = __builtin_fmaf(half_eps, f.a[v][i], p.a[v][i]);  // the next step's first half kick
        }
    }
  }
  return e;
}

// The AUDIT form of the trajectory (ebm_hmc_chain_audit_f32; tests only): the reference's safe-mode leapfrog step literally
// (integrators/leapfrog.py:156-185, restated in oracle/hmc.py) -- the force re-evaluated at the top of every step, two separate
// half kicks, every multiply and add rounded on its own (this file is compiled with -ffp-contract=off and nothing below is an
// explicit FMA), the drift divided by max(m, 1e-10) per step, both nan_to_num_ scrubs on every step.  For the element-wise
// energies the gradient is bit-identical to autograd's (rows.h), so given the same accept decisions the STATE is the reference's
// bit for bit: what the fast body above (merged kicks, FMAs, the hoisted eps / m) trades away is then a measured quantity
// (tests/test_hmc_audit_gpu.py), not a claim.  2 L evaluations per transition + 1 for E(x'), like the reference.
template <bool HAS_MASS, class En, class LaneT>
__device__ __forceinline__ float leapfrog_literal(const En& en, const LaneT& L, Slice<LaneT::NV>& x, Slice<LaneT::NV>& p,
                                                  const Slice<HAS_MASS ? LaneT::NV : 1>& m_clamped, float eps, float half_eps,
                                                  int n_steps) {
  constexpr int NV = LaneT::NV;
  Slice<NV> g;
  for (int l = 0; l < n_steps; ++l) {
    (void)en.template eval<true>(L, x, g);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float force = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);
        const float ph = p.a[v][i] + half_eps * force;          // p + (0.5 eps) * force
        float step = eps * ph;                                   // eps * p_half
        if constexpr (HAS_MASS) step = step / m_clamped.a[v][i]; // ... / max(m, 1e-10)
        p.a[v][i] = ph;
        x.a[v][i] = L.ok(v, i) ? x.a[v][i] + step : 0.0f;
      }
    (void)en.template eval<true>(L, x, g);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float force = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);
        const float pn = p.a[v][i] + half_eps * force;
        x.a[v][i] = nan_to_num0(x.a[v][i]);
        p.a[v][i] = L.ok(v, i) ? nan_to_num0(pn) : 0.0f;
      }
  }
  return en.template eval<true>(L, x, g);  // model(x'), on the scrubbed proposal (samplers/hmc.py:270-274)
}

// MASS: 0 = identity mass (no mass registers at all), 1 = scalar or diagonal mass (three
// per-slot forms kept in VGPRs).  XC_LDS: park the accepted state in a lane-private LDS slot
// while the proposal is integrated (wide rows: frees 4*NV VGPRs).
// DIAG: emit the per-block diagnostics records at the kept transitions (a compile-time switch: the call into
// diag::emit cost the one-lane-per-chain mixture kernel 4 % through register pressure even when never taken).
// CARRY: carry energy and force from transition to transition (see below); false re-evaluates both at the top of
// every transition, the reference's own sequence (kept for A/B measurements).
// LITERAL: the audit form (leapfrog_literal; requires CARRY = false: energy and force re-evaluated at the top of every transition).
template <int KIND, int G, int NV, bool FULL, int MASS, bool DIAG, bool CARRY = true, bool LITERAL = false>
__device__ __forceinline__ void hmc_chain_body(const HmcArgs& a) {
  static_assert(!LITERAL || !CARRY, "the audit form re-evaluates at the top of every transition");
  using LaneT = Lane<G, NV, FULL>;
  constexpr bool XC_LDS = NV >= 4;
  LaneT L;
  L.init(a.n_chains, a.dim);
  const Smem S = carve_smem<NV>(hmc_smem, a.param_floats);
  stage_params(a.energy, a.dim, S.param);
  Energy<KIND, LaneT> en;
  en.init(a.energy, L, S);
  // lane-private parking slots sit behind the parameter / exchange area: [v][thread] float4
  // (indexed straight off the LDS array: a generic pointer into it made hipcc 7.2 emit an illegal
  //  V_CMP_NE_U32 against src_shared_base for the null check of the address-space cast)
  const int park0 = a.park_offset_floats + 4 * (int)threadIdx.x;
  auto park_put = [&](int slot, const float (&q)[4]) {
    *reinterpret_cast<float4*>(&hmc_smem[park0 + slot * (4 * kBlock)]) = make_float4(q[0], q[1], q[2], q[3]);
  };
  auto park_get = [&](int slot, float (&q)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(&hmc_smem[park0 + slot * (4 * kBlock)]);
    q[0] = t.x; q[1] = t.y; q[2] = t.z; q[3] = t.w;
  };

  const int64_t row = L.active ? L.chain * (int64_t)a.dim : 0;
  Slice<NV> xc;  // current (accepted) state
  load_slice(L, a.x, row, xc);

  // mass forms per slot: raw (kinetic energy), sqrt (momentum draw), clamped (drift)
  constexpr bool has_mass = MASS != 0;
  const bool diag_mass = has_mass && a.mass_kind == EBM_MASS_DIAG;
  Slice<has_mass ? NV : 1> m_raw, m_sqrt, m_safe;
  if constexpr (has_mass) {
    if (diag_mass) load_param_slice(L, a.mass_diag, 1.0f, m_raw);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if (diag_mass) {
          // (sqrtf is correctly rounded here -- hipcc's default -fhip-fp32-correctly-rounded-divide-sqrt; torch's CPU sqrt is NOT:
          //  it lands on the wrong side of near-halfway cases, differently on different hosts -- tests/test_hmc_audit_gpu.py)
          m_sqrt.a[v][i] = sqrtf(m_raw.a[v][i]);
          m_safe.a[v][i] = m_raw.a[v][i] < 1e-10f ? 1e-10f : m_raw.a[v][i];
        } else {
          m_raw.a[v][i] = a.mass_raw;
          m_sqrt.a[v][i] = a.mass_sqrt;
          m_safe.a[v][i] = a.mass_safe;
        }
      }
  }

  // K(p) = 0.5 p^T M^-1 p, clamped to [0, 1e10]  (samplers/hmc.py:136-159, :251-254)
  auto kinetic = [&](const Slice<NV>& q) -> float {
    float acc = 0.0f;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float sq = q.a[v][i] * q.a[v][i];
        if constexpr (has_mass) {
          if (diag_mass) sq = sq / m_raw.a[v][i];
        }
        acc += L.ok(v, i) ? sq : 0.0f;
      }
    float k = 0.5f * group_sum<G>(acc);
    if constexpr (has_mass) {
      if (!diag_mass) k = k / a.mass_raw;
    }
    return clamp_nanprop(k, 0.0f, 1e10f);
  };

  const int64_t traj_row = L.active ? L.chain * (int64_t)a.n_kept * a.dim : 0;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  int keep = 0;
  const bool keeping = a.traj != nullptr || DIAG;
  float eps = a.eps;

  // Energy and clamped force of the state the chain holds are CARRIED from transition to transition: an accepted
  // proposal brings its own E1 and end-of-trajectory force (what the reference recomputes at the top of the next
  // transition on the same x: bit-identical), a rejected one keeps the saved pair.  A transition costs L
  // evaluations instead of L + 1.  The pair of the INITIAL state comes out of the same code: the loop starts with
  // a pseudo-transition t = -1 -- zero momentum, zero step size, one leapfrog step (x + 0 * p is x bit for bit,
  // the step's last evaluation is E(x), dE/dx), always "accepted", nothing written -- so the energy is inlined
  // at ONE call site (a second one in front of the loop trips an instruction-selection bug of hipcc 7.2 and costs
  // the register allocation of the hot loop).
  Slice<NV> f;  // clamped force -dE/dx: at the current state between trajectories, then along the trajectory
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int i = 0; i < 4; ++i) f.a[v][i] = 0.0f;
  float e_cur = 0.0f;
  // wide rows: the carried force lives in LDS between trajectories (frees 4*NV VGPRs across the momentum draw and
  // the accept step).  F_TWO (NV == 4): TWO force slots -- the end-of-trajectory force is parked in the spare one
  // right behind the last leapfrog step, before H1 and the accept arithmetic (holding it in registers until the
  // decision costs the element-wise kernels their third wave per SIMD); accepting flips which slot is current.
  constexpr bool F_TWO = CARRY && XC_LDS && NV <= 4;
  int fcur = 0;
  if constexpr (CARRY && XC_LDS) {
#pragma unroll
    for (int v = 0; v < NV; ++v) park_put(NV + v, f.a[v]);
  }

  for (int t = CARRY ? -1 : 0; t < a.n_mh; ++t) {
    const bool init = CARRY && t < 0;
    if constexpr (!CARRY) {  // re-evaluate at the top of every transition (the reference's own sequence)
      e_cur = en.template eval<true>(L, xc, f);
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) f.a[v][i] = clamp_nanprop(-f.a[v][i], -1e6f, 1e6f);
    }
    if (a.eps_table && !init) eps = a.eps_table[t];
    const float eps_t = init ? 0.0f : eps;
    const float half_eps = 0.5f * eps_t;

    // ---- momentum draw: p ~ N(0, M)  (samplers/hmc.py:92-134)
    Slice<NV> p;
    if (init) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) p.a[v][i] = 0.0f;
    } else {
      if (a.p_noise) load_slice(L, a.p_noise, ((int64_t)t * a.n_chains) * a.dim + row, p);
      else normal_slice(L, a.key, a.step0 + 2ull * (uint64_t)t, p);
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          float pv = p.a[v][i];
          if constexpr (has_mass) pv = pv * m_sqrt.a[v][i];
          p.a[v][i] = L.ok(v, i) ? pv : 0.0f;
        }
    }

    // ---- the accept uniform, drawn here (one register across the trajectory) rather than after it: behind the
    //      trajectory the Philox temporaries would sit on top of the live end-of-trajectory force
    float uu;
    if (init) uu = -1.0f;
    else if (a.u) uu = L.active ? a.u[(int64_t)t * a.n_chains + L.chain] : 2.0f;
    else uu = u01_half_open(pick(philox_at(a.key, (uint64_t)L.chain >> 2, a.step0 + 2ull * (uint64_t)t + 1ull),
                                 (int)(L.chain & 3)));

    // ---- H0 from the carried energy
    const float e0 = e_cur;
    const float h0 = clamp_nanprop(e0, -1e10f, 1e10f) + kinetic(p);

    // ---- proposal (integrated in place in xc's registers when the old state is parked in LDS; its force is
    //      parked next to it); narrow rows keep both in registers
    Slice<XC_LDS ? 1 : NV> f_keep;
    if constexpr (XC_LDS) {
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        park_put(v, xc.a[v]);
        if constexpr (CARRY) park_get(NV * (1 + fcur) + v, f.a[v]);
      }
    } else {
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) f_keep.a[v][i] = f.a[v][i];
    }
    Slice<NV> xprop_store;
    Slice<NV>& x = XC_LDS ? xc : xprop_store;
    if constexpr (!XC_LDS) x = xc;
    const int n_lf = init ? 1 : a.n_leapfrog;
    float e1;
    if constexpr (LITERAL) {
      if constexpr (has_mass) e1 = leapfrog_literal<true>(en, L, x, p, m_safe, eps_t, half_eps, n_lf);
      else e1 = leapfrog_literal<false>(en, L, x, p, m_safe, eps_t, half_eps, n_lf);
    } else if constexpr (has_mass) {
      // drift x += eps * p / max(m, 1e-10): the quotient eps / m is formed once per transition (an IEEE
      // division per coordinate per LEAPFROG STEP cost 40 % of the massed kernel), the step is one FMA
      Slice<NV> drift_scale;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) drift_scale.a[v][i] = eps_t / m_safe.a[v][i];
      e1 = leapfrog_steps<true>(en, L, x, p, f, drift_scale, eps_t, half_eps, n_lf, e0, init);
    }
    else e1 = leapfrog_steps<false>(en, L, x, p, f, x, eps_t, half_eps, n_lf, e0, init);
    if constexpr (F_TWO) {
#pragma unroll
      for (int v = 0; v < NV; ++v) park_put(NV * (2 - fcur) + v, f.a[v]);
    }
    const float h1 = clamp_nanprop(e1, -1e10f, 1e10f) + kinetic(p);

    // ---- Metropolis accept (samplers/hmc.py:277-292)
    const float dlt = clamp_nanprop(h0 - h1, -50.0f, 50.0f);
    float acc_p = expf(dlt);
    acc_p = (acc_p > 1.0f) ? 1.0f : acc_p;  // clamp_(max=1); NaN stays NaN and rejects
    // (the pseudo-transition always takes its "proposal": the unchanged state with its energy and force)
    const bool accept = init || (L.active && (uu < acc_p));
    if (accept) e_cur = e1;
    if constexpr (XC_LDS) {
      if (accept) {  // the proposal's force becomes the carried one
        if constexpr (F_TWO) {
          fcur ^= 1;
        } else if constexpr (CARRY) {
#pragma unroll
          for (int v = 0; v < NV; ++v) park_put(NV + v, f.a[v]);
        }
      } else {       // rejected: bring the parked state back (its force is still parked)
#pragma unroll
        for (int v = 0; v < NV; ++v) park_get(v, xc.a[v]);
      }
    } else {
      if (accept) {
        xc = x;
      } else {
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
          for (int i = 0; i < 4; ++i) f.a[v][i] = f_keep.a[v][i];
      }
    }
    if (init) continue;

    const bool leader = L.active && L.lg == 0;
    if (a.accept_mask && leader) a.accept_mask[(int64_t)t * a.n_chains + L.chain] = accept ? 1 : 0;
    if (a.accept_count) {  // wavefront-level count, one atomic per wave
      const unsigned long long b = __ballot(accept && leader);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(a.accept_count + t, (uint32_t)__popcll(b));
    }

    if (keeping && --until_keep == 0) {
      until_keep = a.thin;
      if (a.traj) {
        store_slice(L, a.traj, traj_row + keep_off, xc);
        keep_off += a.dim;
      }
      if constexpr (DIAG) {
        // samplers/hmc.py:294-310: population mean / var, mean of the clamped energy of the state the chain
        // holds now (its accepted proposal's E1, else E0 -- what the reference re-evaluates), acceptance rate
        // The tile: with XC_LDS the slot the old state was parked in -- dead until the next transition parks
        // again, and exactly one block of rows wide -- so the records cost no LDS beyond the scratch rows (a tile of
        // its own took the one-lane-per-chain kernels from two workgroups per CU to one: +25 % on config 3).  The
        // barrier: a slower wave may still have to bring its parked state back.
        float* scratch = hmc_smem + a.diag_offset_floats;
        float* tile = scratch + diag::scratch_floats(a.diag.S);
        if constexpr (XC_LDS) {
          tile = hmc_smem + a.park_offset_floats;
          __syncthreads();
        }
        tile_store(L, tile, xc);
        const float e_now = clamp_nanprop(e_cur, -1e10f, 1e10f);
        diag::emit(a.diag, keep, tile, scratch, tile_valid<G>(a.n_chains, a.dim), a.dim, leader ? e_now : 0.0f,
                   (accept && leader) ? 1.0f : 0.0f);
        ++keep;
      }
    }
  }
  store_slice(L, a.x, row, xc);
}

template <int KIND, int G, int NV, bool FULL, int MASS, bool DIAG>
__global__ __launch_bounds__(kBlock) void hmc_chain_kernel(HmcArgs a) {
  hmc_chain_body<KIND, G, NV, FULL, MASS, DIAG>(a);
}

// The audit instantiation: one vector per lane, the plain geometry of pick_geometry (dims up to 256), no records.
template <int KIND, int G, int MASS>
__global__ __launch_bounds__(kBlock) void hmc_chain_kernel_literal(HmcArgs a) {
  hmc_chain_body<KIND, G, 1, false, MASS, false, false, true>(a);
}

template <int KIND>
void launch_literal(const Geometry& geo, dim3 grid, size_t smem, hipStream_t st, const HmcArgs& a) {
  const dim3 block(kBlock);
#define EBM_HMC_LIT(GV)                                                                                       \
  case GV:                                                                                                    \
    if (a.mass_kind == EBM_MASS_NONE) hipLaunchKernelGGL((hmc_chain_kernel_literal<KIND, GV, 0>), grid, block, smem, st, a); \
    else hipLaunchKernelGGL((hmc_chain_kernel_literal<KIND, GV, 1>), grid, block, smem, st, a);              \
    break;
  switch (geo.G) {
    EBM_HMC_LIT(1) EBM_HMC_LIT(2) EBM_HMC_LIT(4) EBM_HMC_LIT(8) EBM_HMC_LIT(16) EBM_HMC_LIT(32)
    default:
      if (a.mass_kind == EBM_MASS_NONE) hipLaunchKernelGGL((hmc_chain_kernel_literal<KIND, 64, 0>), grid, block, smem, st, a);
      else hipLaunchKernelGGL((hmc_chain_kernel_literal<KIND, 64, 1>), grid, block, smem, st, a);
  }
#undef EBM_HMC_LIT
}

// Same body held to 256 VGPRs (two waves per SIMD).  For the one-lane-per-chain mixture kernel:
// its leapfrog loop fits, only the cold paths (prologue, large-K fallback, scrub) spill, and the
// second wave is worth 1.56 -> 1.22 ms on BASELINE config 3.  (A template-dependent expression in
// __launch_bounds__ is silently ignored by hipcc 7.2, hence the second entry point.)
// Measured and NOT taken (round 2, profiles/r02_hmc_c3_experiments.txt): three waves per SIMD at 168 VGPRs without
// the carried force (1.06 ms per 10 transitions against 1.00: the cold-path spills grow and the third wave does
// not buy back the scalar-load waits), and two waves without the carried force (1.05).
// For the mixture this kernel looks at the active-column mask first: means that differ in columns 0..3 only take the
// body built on Energy<kGmmSlot1> (rows.h) -- a wave-uniform branch, both bodies share the register budget.
template <int KIND, int G, int NV, bool FULL, int MASS, bool DIAG>
__global__ __launch_bounds__(kBlock, 2) void hmc_chain_kernel_w2(HmcArgs a) {
  if constexpr (KIND == EBM_ENERGY_GMM && G == 1 && FULL && NV >= 4) {
    if (gmm_is_slot1(a.energy)) {
      hmc_chain_body<kGmmSlot1, G, NV, FULL, MASS, DIAG, true>(a);
      return;
    }
  }
  hmc_chain_body<KIND, G, NV, FULL, MASS, DIAG, true>(a);
}

// KERNEL<KIND, G, NV, FULL, MASS, DIAG> over the runtime geometry (see rows.h: EBM_GEO_LAUNCH)
template <int KIND, int MASS, bool DIAG>
void launch_geo(const Geometry& geo, dim3 grid, size_t smem, hipStream_t st, const HmcArgs& a) {
  const dim3 block(kBlock);
#define EBM_HMC_G(GV, NVV, FULLV) hipLaunchKernelGGL((hmc_chain_kernel<KIND, GV, NVV, FULLV, MASS, DIAG>), grid, block, smem, st, a)
  if (geo.NV == 1) {
    switch (geo.G) {
      case 1:  if (geo.full) EBM_HMC_G(1, 1, true);  else EBM_HMC_G(1, 1, false);  break;
      case 2:  if (geo.full) EBM_HMC_G(2, 1, true);  else EBM_HMC_G(2, 1, false);  break;
      case 4:  if (geo.full) EBM_HMC_G(4, 1, true);  else EBM_HMC_G(4, 1, false);  break;
      case 8:  if (geo.full) EBM_HMC_G(8, 1, true);  else EBM_HMC_G(8, 1, false);  break;
      case 16: if (geo.full) EBM_HMC_G(16, 1, true); else EBM_HMC_G(16, 1, false); break;
      case 32: if (geo.full) EBM_HMC_G(32, 1, true); else EBM_HMC_G(32, 1, false); break;
      default: if (geo.full) EBM_HMC_G(64, 1, true); else EBM_HMC_G(64, 1, false); break;
    }
  } else if (geo.NV == 3) {  // element-wise energies, row widths in (2^k, 1.5 2^k] vectors: three vectors per lane (hmc.hip: hmc_geometry)
    if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL || KIND == EBM_ENERGY_HARMONIC) {
      switch (geo.G) {
        case 1:  EBM_HMC_G(1, 3, false);  break;
        case 2:  EBM_HMC_G(2, 3, false);  break;
        case 4:  EBM_HMC_G(4, 3, false);  break;
        case 8:  EBM_HMC_G(8, 3, false);  break;
        case 16: EBM_HMC_G(16, 3, false); break;
        case 32: EBM_HMC_G(32, 3, false); break;
        default: EBM_HMC_G(64, 3, false); break;
      }
    }
  } else if (geo.G == 64 && geo.NV == 2) {
    // (element-wise energies at exactly 512 / 1024 dims: the full-row form -- no per-element masks)
    if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL || KIND == EBM_ENERGY_HARMONIC) {
      if (geo.full) EBM_HMC_G(64, 2, true);
      else EBM_HMC_G(64, 2, false);
    } else {
      EBM_HMC_G(64, 2, false);
    }
  } else if (geo.G == 64 && geo.NV == 4) {
    if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL || KIND == EBM_ENERGY_HARMONIC) {
      if (geo.full) EBM_HMC_G(64, 4, true);
      else EBM_HMC_G(64, 4, false);
    } else {
      EBM_HMC_G(64, 4, false);
    }
  } else if (geo.NV == 4 && geo.full && (geo.G == 4 || geo.G == 8)) {  // element-wise energies at dim 64 / 128
    if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL || KIND == EBM_ENERGY_HARMONIC) {
      if (geo.G == 4) EBM_HMC_G(4, 4, true);
      else EBM_HMC_G(8, 4, true);
    }
  } else if (geo.G == 4 && geo.NV == 2) {  // dim-32 alternatives (full rows only)
    EBM_HMC_G(4, 2, true);
  } else if (geo.G == 2 && geo.NV == 4) {
    EBM_HMC_G(2, 4, true);
  } else if constexpr (KIND == EBM_ENERGY_GMM) {
    // (1, 8) is the small-mixture geometry (K <= 8 at dim 32, hmc.hip: hmc_geometry); with identity mass those calls never get
    // here -- hmc_ring.hip / hmc_gmm32.hip serve them, records included -- so only the massed form is instantiated
    if constexpr (MASS != 0) hipLaunchKernelGGL((hmc_chain_kernel_w2<KIND, 1, 8, true, MASS, DIAG>), grid, block, smem, st, a);
  } else {
    EBM_HMC_G(1, 8, true);
  }
#undef EBM_HMC_G
}

template <int KIND, bool DIAG>
void launch_kind(const Geometry& geo, dim3 grid, size_t smem, hipStream_t st, const HmcArgs& a) {
  if (a.mass_kind == EBM_MASS_NONE) launch_geo<KIND, 0, DIAG>(geo, grid, smem, st, a);
  else launch_geo<KIND, 1, DIAG>(geo, grid, smem, st, a);
}

}  // namespace hmc
}  // namespace ebm
