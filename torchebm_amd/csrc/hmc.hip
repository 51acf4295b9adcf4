// Fused HMC transition kernel for gfx950: momentum draw, Hamiltonian, L safe-mode leapfrog
// steps and the Metropolis accept for n_mh transitions in ONE launch, state in VGPRs.
// Layout and energies: rows.h.  Reference: torchebm/samplers/hmc.py:243-312,
// torchebm/integrators/leapfrog.py:156-185, torchebm/core/base_integrator.py:875-889.
#include <cstdlib>

#include "rows.h"

namespace ebm {
using namespace rows;

namespace {

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
};

extern __shared__ __attribute__((aligned(16))) float hmc_smem[];

// L leapfrog steps in safe mode.  On entry f = clamp(-dE/dx) at x; on exit x, p are the
// proposal, f the clamped force there, and the return value is E(x).
//  * The clamped force at the end of a step is bit-identical to the one the reference
//    recomputes at the start of the next step, so it is carried over.
//  * torch's nan_to_num_ is the identity on finite values: the common path only *tests*
//    x, p for non-finite values (one v_cmp_class each); the scrub itself, and the force
//    re-evaluation the reference then performs on the scrubbed x, run only for lane groups
//    that actually hold a NaN/inf.
template <bool HAS_MASS, class En, class LaneT>
__device__ __forceinline__ float leapfrog_steps(const En& en, const LaneT& L, Slice<LaneT::NV>& x,
                                                Slice<LaneT::NV>& p, Slice<LaneT::NV>& f,
                                                const Slice<LaneT::NV>& m_safe, float eps, float half_eps,
                                                int n_steps, float e_in) {
  constexpr int NV = LaneT::NV;
  float e = e_in;
  for (int l = 0; l < n_steps; ++l) {
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float ph = p.a[v][i] + half_eps * f.a[v][i];
        float step = eps * ph;
        if constexpr (HAS_MASS) step = step / m_safe.a[v][i];
        p.a[v][i] = ph;
        x.a[v][i] = L.ok(v, i) ? x.a[v][i] + step : 0.0f;
      }
    Slice<NV> g;
    e = en.template eval<true>(L, x, g);
    bool bad = false;
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float fn = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);
        const float pn = p.a[v][i] + half_eps * fn;
        f.a[v][i] = fn;
        p.a[v][i] = L.ok(v, i) ? pn : 0.0f;
        bad |= !__builtin_isfinite(pn) | !__builtin_isfinite(x.a[v][i]);
      }
    if (group_any<LaneT::G>(bad)) {  // rare: scrub, then re-evaluate on the scrubbed position
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          p.a[v][i] = nan_to_num0(p.a[v][i]);
          x.a[v][i] = nan_to_num0(x.a[v][i]);
        }
      e = en.template eval<true>(L, x, g);
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) f.a[v][i] = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);
    }
  }
  return e;
}

template <int KIND, int G, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void hmc_chain_kernel(HmcArgs a) {
  using LaneT = Lane<G, NV, FULL>;
  LaneT L;
  L.init(a.n_chains, a.dim);
  const Smem S = carve_smem<NV>(hmc_smem, a.param_floats);
  stage_params(a.energy, a.dim, S.param);
  Energy<KIND, LaneT> en;
  en.init(a.energy, L, S);

  const int64_t row = L.active ? L.chain * (int64_t)a.dim : 0;
  Slice<NV> xc;  // current (accepted) state
  load_slice(L, a.x, row, xc);

  // mass forms per slot: raw (kinetic energy), sqrt (momentum draw), clamped (drift)
  const bool has_mass = a.mass_kind != EBM_MASS_NONE;
  const bool diag_mass = a.mass_kind == EBM_MASS_DIAG;
  Slice<NV> m_raw, m_sqrt, m_safe;
  if (diag_mass) load_param_slice(L, a.mass_diag, 1.0f, m_raw);
#pragma unroll
  for (int v = 0; v < NV; ++v)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (diag_mass) {
        m_sqrt.a[v][i] = sqrtf(m_raw.a[v][i]);
        m_safe.a[v][i] = m_raw.a[v][i] < 1e-10f ? 1e-10f : m_raw.a[v][i];
      } else {
        m_raw.a[v][i] = a.mass_raw;
        m_sqrt.a[v][i] = a.mass_sqrt;
        m_safe.a[v][i] = a.mass_safe;
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
        if (diag_mass) sq = sq / m_raw.a[v][i];
        acc += L.ok(v, i) ? sq : 0.0f;
      }
    float k = 0.5f * group_sum<G>(acc);
    if (has_mass && !diag_mass) k = k / a.mass_raw;
    return clamp_nanprop(k, 0.0f, 1e10f);
  };

  const int64_t traj_row = L.active ? L.chain * (int64_t)a.n_kept * a.dim : 0;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  float eps = a.eps;

  for (int t = 0; t < a.n_mh; ++t) {
    if (a.eps_table) eps = a.eps_table[t];
    const float half_eps = 0.5f * eps;

    // ---- momentum draw: p ~ N(0, M)  (samplers/hmc.py:92-134)
    Slice<NV> p;
    if (a.p_noise) load_slice(L, a.p_noise, ((int64_t)t * a.n_chains) * a.dim + row, p);
    else normal_slice(L, a.key, a.step0 + 2ull * (uint64_t)t, p);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        float pv = p.a[v][i];
        if (has_mass) pv = pv * m_sqrt.a[v][i];
        p.a[v][i] = L.ok(v, i) ? pv : 0.0f;
      }

    // ---- H0 and the first (clamped) force
    Slice<NV> g;
    const float e0 = en.template eval<true>(L, xc, g);
    const float h0 = clamp_nanprop(e0, -1e10f, 1e10f) + kinetic(p);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) g.a[v][i] = clamp_nanprop(-g.a[v][i], -1e6f, 1e6f);

    // ---- proposal
    Slice<NV> x = xc;
    float e1;
    if (has_mass) e1 = leapfrog_steps<true>(en, L, x, p, g, m_safe, eps, half_eps, a.n_leapfrog, e0);
    else e1 = leapfrog_steps<false>(en, L, x, p, g, m_safe, eps, half_eps, a.n_leapfrog, e0);
    const float h1 = clamp_nanprop(e1, -1e10f, 1e10f) + kinetic(p);

    // ---- Metropolis accept (samplers/hmc.py:277-292)
    const float dlt = clamp_nanprop(h0 - h1, -50.0f, 50.0f);
    float acc_p = expf(dlt);
    acc_p = (acc_p > 1.0f) ? 1.0f : acc_p;  // clamp_(max=1); NaN stays NaN and rejects
    float uu;
    if (a.u) uu = L.active ? a.u[(int64_t)t * a.n_chains + L.chain] : 2.0f;
    else uu = u01_half_open(pick(philox_at(a.key, (uint64_t)L.chain >> 2, a.step0 + 2ull * (uint64_t)t + 1ull),
                                 (int)(L.chain & 3)));
    const bool accept = L.active && (uu < acc_p);
    if (accept) xc = x;

    const bool leader = L.active && L.lg == 0;
    if (a.accept_mask && leader) a.accept_mask[(int64_t)t * a.n_chains + L.chain] = accept ? 1 : 0;
    if (a.accept_count) {  // wavefront-level count, one atomic per wave
      const unsigned long long b = __ballot(accept && leader);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(a.accept_count + t, (uint32_t)__popcll(b));
    }

    if (a.traj && --until_keep == 0) {
      until_keep = a.thin;
      store_slice(L, a.traj, traj_row + keep_off, xc);
      keep_off += a.dim;
    }
  }
  store_slice(L, a.x, row, xc);
}

}  // namespace

int launch_hmc_chain(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh,
                     int32_t n_leapfrog, float eps, const float* eps_table, int32_t mass_kind,
                     double mass_scalar, const float* mass_diag, int32_t thin, float* traj,
                     uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise,
                     const float* u, uint64_t seed, uint64_t offset, hipStream_t st) {
  Geometry geo;
  if (!pick_geometry(dim, geo)) return fail(EBM_EDIM, "ebm_hmc_chain_f32: dim %d > 1024 is not supported by the fused kernel", dim);
  HmcArgs a;
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.n_mh = n_mh; a.n_leapfrog = n_leapfrog;
  a.eps = eps; a.eps_table = eps_table; a.mass_kind = mass_kind;
  a.mass_raw = (float)mass_scalar;
  a.mass_sqrt = (float)sqrt(mass_scalar);
  a.mass_safe = (float)(mass_scalar < 1e-10 ? 1e-10 : mass_scalar);
  a.mass_diag = mass_diag; a.thin = thin; a.n_kept = n_mh / thin; a.traj = traj;
  a.accept_mask = accept_mask; a.accept_count = accept_count; a.p_noise = p_noise; a.u = u;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  size_t smem = 0;
  plan_params(e, dim, geo, a.energy, a.param_floats, smem);
  const int64_t blocks = blocks_for(n_chains, geo);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_hmc_chain_f32: too many chains for one launch");
  // experiment hook: EBM_HMC_NV=2|4|8 re-shapes a dim-32 row to (G, NV) = (4,2) | (2,4) | (1,8)
  static const int nv_env = [] {
    const char* s = getenv("EBM_HMC_NV");
    return s ? atoi(s) : 0;
  }();
  if (dim == 32 && (nv_env == 2 || nv_env == 4 || nv_env == 8) &&
      (e.kind == EBM_ENERGY_GMM || e.kind == EBM_ENERGY_DOUBLE_WELL) && e.n_comp <= 8) {
    const int g = 8 / nv_env;
    Geometry alt{g, nv_env, true};
    plan_params(e, dim, alt, a.energy, a.param_floats, smem);
    const dim3 grid((unsigned)blocks_for(n_chains, alt)), block(kBlock);
#define EBM_ALT(KIND)                                                                                     \
  do {                                                                                                    \
    if (nv_env == 2) hipLaunchKernelGGL((hmc_chain_kernel<KIND, 4, 2, true>), grid, block, smem, st, a);      \
    else if (nv_env == 4) hipLaunchKernelGGL((hmc_chain_kernel<KIND, 2, 4, true>), grid, block, smem, st, a); \
    else hipLaunchKernelGGL((hmc_chain_kernel<KIND, 1, 8, true>), grid, block, smem, st, a);                  \
  } while (0)
    if (e.kind == EBM_ENERGY_GMM) EBM_ALT(EBM_ENERGY_GMM);
    else EBM_ALT(EBM_ENERGY_DOUBLE_WELL);
#undef EBM_ALT
    return check_launch("ebm_hmc_chain_f32");
  }
  EBM_KIND_LAUNCH(hmc_chain_kernel, e.kind, geo, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_hmc_chain_f32");
}

}  // namespace ebm
