// HMC transition kernel for mixtures whose component means differ in columns 0..3 only (the `aux` mask of
// EBM_ENERGY_GMM equal to 1: a K-mode mixture of a plane embedded in a wider state -- BASELINE config 3's
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
//   * the 28 shared columns are held as y = x - mu_0 (exact when mu_0 is zero there, as on the ring; one rounding at
//     load and one at store otherwise -- the HMC state is a tolerance tier), so their force is ONE packed multiply,
//     f = -y / sigma^2, never stored: the kick consumes it on the spot.  Only the four active-slot forces are carried
//     from step to step (and from transition to transition: an accepted proposal's end-of-trajectory force starts
//     the next trajectory);
//   * the accepted state is parked in a lane-private LDS slot during the proposal (32 KiB per workgroup, four
//     workgroups per CU), the carried active force next to it;
//   * every constant of the step loop is a scalar register: the 8 x 4 active means pre-scaled by log2(e) / sigma^2
//     (the logits come out in base 2: v_exp_f32 directly) and the logit offsets;
//   * safe mode costs ONE running NaN-propagating maximum: m = maximum3(m, |f_a|, |f_b|) per packed pair.  The
//     reference clamps the force to +-1e6 and scrubs non-finite x / p after every step; both are the identity while
//     every |f| <= 1e6 and p starts below 1e30 (the force is clamped, so p cannot leave the finite range within a
//     trajectory), and a non-finite coordinate or logit makes a force NaN / inf.  A chain whose trajectory ends with
//     !(m <= 1e6) is REDONE from the parked state by the literal sequence (NaN-propagating clamps, half kicks,
//     scrub after every step, force re-evaluated on the scrubbed state) -- cold code behind a wave-level branch.
// Per leapfrog step and chain: ~140 VALU instructions (~200 issue units) where the shared body took ~220 (~300).
#include "hmc_kernel.h"

namespace ebm {
namespace hmc {

typedef float v2f __attribute__((ext_vector_type(2)));

namespace {

__device__ __forceinline__ float to_sgpr(float v) {
  return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}
// NaN-propagating maximum of three (v_maximum3_f32)
__device__ __forceinline__ float max3np(float a, float b, float c) {
  return __builtin_elementwise_maximum(__builtin_elementwise_maximum(a, b), c);
}
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f splat(float v) { return v2f{v, v}; }

constexpr float kLog2e = 1.44269504088896340736f;
constexpr float kLn2 = 0.69314718055994530942f;
constexpr int kTabFloats = 32 + 8 + 32;  // raw active means [8][4], log-weights [8], row 0 of the means [32]

}  // namespace

template <int NV>
__global__ __launch_bounds__(kBlock, 4) void hmc_slot1_kernel(HmcArgs a) {
  if (!gmm_is_slot1(a.energy)) return;  // any other mask: the dense kernel, launched behind this one, does the work
  static_assert(NV == 8, "dim 32");
  constexpr int D = 4 * NV, NP = 2 * NV;  // columns, packed pairs; pairs 0 and 1 are the active slot
  using LaneT = Lane<1, NV, true>;
  LaneT L;
  L.init(a.n_chains, a.dim);

  // ---- LDS: [table | parked state, [v][thread] float4 | parked active force, [thread] float4]
  float* const tab = hmc_smem;
  const int K = a.energy.n_comp;
  for (int i = threadIdx.x; i < kTabFloats; i += kBlock) {
    float v;
    if (i < 32) {
      const int k = i >> 2, kk = k < K ? k : K - 1;
      v = a.energy.dev0[kk * D + (i & 3)];
    } else if (i < 40) {
      v = (i - 32) < K ? a.energy.dev1[i - 32] : -__builtin_inff();
    } else {
      v = a.energy.dev0[i - 40];
    }
    tab[i] = v;
  }
  const int park0 = kTabFloats + 4 * (int)threadIdx.x;
  const int fpark = kTabFloats + NV * 4 * kBlock + 4 * (int)threadIdx.x;
  __syncthreads();

  // ---- constants of the step loop (wave-uniform: scalar registers)
  const float invs2 = a.energy.s1, inv2s2 = a.energy.s0;
  float m2[8][4];  // mu_k[i] * log2(e) / sigma^2
  float c2v[8];    // (logw_k - |mu_k[0:4]|^2 / (2 sigma^2)) * log2(e): the x-independent part of the base-2 logit
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    float nrm = 0.0f;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float mu = tab[4 * k + i];
      nrm = __builtin_fmaf(mu, mu, nrm);
      m2[k][i] = to_sgpr(mu * (invs2 * kLog2e));
    }
    c2v[k] = __builtin_fmaf(-nrm, inv2s2, tab[32 + k]) * kLog2e;
  }
#pragma unroll
  for (int k = 0; k < 8; ++k) c2v[k] = to_sgpr(c2v[k]);
  const v2f NI = splat(-invs2);

  // active-slot force -dE/dx[0:4] at (xa, xb):  f = (sum_k r_k mu_k - x) / sigma^2,  r = softmax of the logits
  auto active_force = [&](v2f xa, v2f xb, const v2f (&C2)[4], float (&fa)[4]) {
    v2f lp[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      v2f t = pk_fma(splat(xa.x), v2f{m2[2 * q][0], m2[2 * q + 1][0]}, C2[q]);
      t = pk_fma(splat(xa.y), v2f{m2[2 * q][1], m2[2 * q + 1][1]}, t);
      t = pk_fma(splat(xb.x), v2f{m2[2 * q][2], m2[2 * q + 1][2]}, t);
      lp[q] = pk_fma(splat(xb.y), v2f{m2[2 * q][3], m2[2 * q + 1][3]}, t);
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
    const v2f ta = xa * NI, tb = xb * NI;
    const float tx[4] = {ta.x, ta.y, tb.x, tb.y};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v2f acc = w[0] * v2f{m2[0][i], m2[1][i]};
#pragma unroll
      for (int q = 1; q < 4; ++q) acc = pk_fma(w[q], v2f{m2[2 * q][i], m2[2 * q + 1][i]}, acc);
      fa[i] = __builtin_fmaf(acc.x + acc.y, s, tx[i]);
    }
  };

  // E(x) in the reference's difference form (the form the shared body evaluates H0 / H1 in):
  //   sum_{d >= 4} (x_d - mu_0d)^2 / (2 sigma^2) - logsumexp_k(logw_k - |x[0:4] - mu_k[0:4]|^2 / (2 sigma^2))
  auto energy_exact = [&](const v2f (&X)[NP]) -> float {
    float logit[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 m = *reinterpret_cast<const float4*>(&tab[4 * k]);
      const v2f da = X[0] - v2f{m.x, m.y}, db = X[1] - v2f{m.z, m.w};
      const v2f d2 = pk_fma(db, db, da * da);
      logit[k] = __builtin_fmaf(-(d2.x + d2.y), inv2s2, tab[32 + k]);
    }
    float top = logit[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) top = __builtin_fmaxf(top, logit[k]);
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += __expf(logit[k] - top);
    v2f sq = {0.0f, 0.0f}, sq_b = {0.0f, 0.0f};
#pragma unroll
    for (int v = 1; v < NV; ++v) {  // the shared columns are held as x - mu_0
      sq = pk_fma(X[2 * v], X[2 * v], sq);
      sq_b = pk_fma(X[2 * v + 1], X[2 * v + 1], sq_b);
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

  const int64_t row = L.active ? L.chain * (int64_t)D : 0;
  // full rows, 16-byte aligned: whole float4 accesses, lanes past the last chain hold zeros and never store
  auto load_row = [&](const float* __restrict__ src, v2f (&R)[NP]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (L.active) q = *reinterpret_cast<const float4*>(src + 4 * v);
      R[2 * v] = v2f{q.x, q.y};
      R[2 * v + 1] = v2f{q.z, q.w};
    }
  };
  // shared columns <-> y = x - mu_0 (row 0 of the means, from the LDS table)
  auto shift = [&](v2f (&R)[NP], float sign) {
#pragma unroll
    for (int v = 1; v < NV; ++v) {
      const float4 mu = *reinterpret_cast<const float4*>(&tab[40 + 4 * v]);
      R[2 * v] = pk_fma(splat(sign), v2f{mu.x, mu.y}, R[2 * v]);      // sign = +-1: the product is exact
      R[2 * v + 1] = pk_fma(splat(sign), v2f{mu.z, mu.w}, R[2 * v + 1]);
    }
  };
  auto store_row = [&](float* __restrict__ dst, const v2f (&R)[NP]) {
    v2f O[NP];
#pragma unroll
    for (int j = 0; j < NP; ++j) O[j] = R[j];
    shift(O, 1.0f);
    if (L.active) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
        *reinterpret_cast<float4*>(dst + 4 * v) = make_float4(O[2 * v].x, O[2 * v].y, O[2 * v + 1].x, O[2 * v + 1].y);
    }
  };
  v2f X[NP];
  load_row(a.x + row, X);
  shift(X, -1.0f);
  auto draw_momentum = [&](int t, v2f (&P)[NP]) {
    if (a.p_noise) {
      load_row(a.p_noise + ((int64_t)t * a.n_chains) * D + row, P);
    } else {
      Slice<NV> s;
      normal_slice(L, a.key, a.step0 + 2ull * (uint64_t)t, s);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        P[2 * v] = v2f{s.a[v][0], s.a[v][1]};
        P[2 * v + 1] = v2f{s.a[v][2], s.a[v][3]};
      }
    }
  };

  const int64_t traj_row = L.active ? L.chain * (int64_t)a.n_kept * D : 0;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  float eps = a.eps;
  float e_cur = 0.0f;
  float fa[4] = {0.0f, 0.0f, 0.0f, 0.0f};  // active-slot force at the state the chain holds

  // t = -1 is a pseudo-transition (zero momentum, zero step size, one leapfrog step, always "accepted", nothing
  // written): x + 0 * p is x bit for bit, so it leaves the energy and the carried force of the initial state.
  for (int t = -1; t < a.n_mh; ++t) {
    const bool init = t < 0;
    if (a.eps_table && !init) eps = a.eps_table[t];
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
    else if (a.u) uu = L.active ? a.u[(int64_t)t * a.n_chains + L.chain] : 2.0f;
    else uu = u01_half_open(pick(philox_at(a.key, (uint64_t)L.chain >> 2, a.step0 + 2ull * (uint64_t)t + 1ull),
                                 (int)(L.chain & 3)));
    const float h0 = clamp_nanprop(e_cur, -1e10f, 1e10f) + kinetic(P);

    // park the accepted state and its active force
#pragma unroll
    for (int v = 0; v < NV; ++v)
      *reinterpret_cast<float4*>(&hmc_smem[park0 + v * (4 * kBlock)]) = make_float4(X[2 * v].x, X[2 * v].y, X[2 * v + 1].x, X[2 * v + 1].y);
    *reinterpret_cast<float4*>(&hmc_smem[fpark]) = make_float4(fa[0], fa[1], fa[2], fa[3]);

    // the logit offsets as vector-register pairs for the trajectory (the packed FMA that adds them already takes the
    // means from scalar registers); made here so that they are not live across the momentum draw
    v2f C2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      C2[q] = v2f{c2v[2 * q], c2v[2 * q + 1]};
      asm volatile("" : "+v"(C2[q]));
    }

    // ---- the trajectory, common path.  m: running NaN-propagating maximum of |force| and of the entry momentum / 1e24
    float m = 0.0f;
#pragma unroll
    for (int j = 0; j < NP; ++j) m = max3np(m, __builtin_fabsf(P[j].x) * 1e-24f, __builtin_fabsf(P[j].y) * 1e-24f);  // |p| >= 1e30 <=> > 1e6
    {
      const v2f H2 = splat(half_eps);
      m = max3np(m, __builtin_fabsf(fa[0]), __builtin_fabsf(fa[1]));
      m = max3np(m, __builtin_fabsf(fa[2]), __builtin_fabsf(fa[3]));
      P[0] = pk_fma(H2, v2f{fa[0], fa[1]}, P[0]);
      P[1] = pk_fma(H2, v2f{fa[2], fa[3]}, P[1]);
#pragma unroll
      for (int j = 2; j < NP; ++j) {
        const v2f F = X[j] * NI;
        m = max3np(m, __builtin_fabsf(F.x), __builtin_fabsf(F.y));
        P[j] = pk_fma(H2, F, P[j]);
      }
    }
    const v2f E2 = splat(eps_t);
    for (int l = 0; l < n_lf; ++l) {
      const v2f K2 = splat(l + 1 >= n_lf ? half_eps : eps_t);  // the next step's first half kick rides along
#pragma unroll
      for (int j = 0; j < NP; ++j) X[j] = pk_fma(E2, P[j], X[j]);
      active_force(X[0], X[1], C2, fa);
      m = max3np(m, __builtin_fabsf(fa[0]), __builtin_fabsf(fa[1]));
      m = max3np(m, __builtin_fabsf(fa[2]), __builtin_fabsf(fa[3]));
      P[0] = pk_fma(K2, v2f{fa[0], fa[1]}, P[0]);
      P[1] = pk_fma(K2, v2f{fa[2], fa[3]}, P[1]);
#pragma unroll
      for (int j = 2; j < NP; ++j) {
        const v2f F = X[j] * NI;
        m = max3np(m, __builtin_fabsf(F.x), __builtin_fabsf(F.y));
        P[j] = pk_fma(K2, F, P[j]);
      }
    }
    float e1 = energy_exact(X);
    // Anything the reference's safe mode would have touched?  (a clamped force, a non-finite coordinate / logit /
    // momentum, a non-finite energy at the end: leapfrog.py:165-185 clamps and scrubs.)
    const bool bad = !init && (!(m <= 1e6f) || !(__builtin_fabsf(e1) < __builtin_inff()));
    if (__builtin_expect(bad, 0)) {
      // ---- the literal sequence from the parked state:  per step  f = clamp(-dE/dx(x)); p += eps/2 f; x += eps p;
      //      f' = clamp(-dE/dx(x)); p += eps/2 f'; scrub x, p
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float4 q = *reinterpret_cast<const float4*>(&hmc_smem[park0 + v * (4 * kBlock)]);
        X[2 * v] = v2f{q.x, q.y};
        X[2 * v + 1] = v2f{q.z, q.w};
      }
      draw_momentum(t, P);
      auto half_kick = [&]() {
        float g[4];
        active_force(X[0], X[1], C2, g);
        P[0].x = __builtin_fmaf(half_eps, clamp_nanprop(g[0], -1e6f, 1e6f), P[0].x);
        P[0].y = __builtin_fmaf(half_eps, clamp_nanprop(g[1], -1e6f, 1e6f), P[0].y);
        P[1].x = __builtin_fmaf(half_eps, clamp_nanprop(g[2], -1e6f, 1e6f), P[1].x);
        P[1].y = __builtin_fmaf(half_eps, clamp_nanprop(g[3], -1e6f, 1e6f), P[1].y);
#pragma unroll
        for (int j = 2; j < NP; ++j) {
          const v2f F = X[j] * NI;
          P[j].x = __builtin_fmaf(half_eps, clamp_nanprop(F.x, -1e6f, 1e6f), P[j].x);
          P[j].y = __builtin_fmaf(half_eps, clamp_nanprop(F.y, -1e6f, 1e6f), P[j].y);
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
#pragma unroll
        for (int j = 0; j < NP; ++j) {
          X[j] = v2f{nan_to_num0(X[j].x), nan_to_num0(X[j].y)};
          P[j] = v2f{nan_to_num0(P[j].x), nan_to_num0(P[j].y)};
        }
      }
      // the force the next trajectory starts from (the reference re-evaluates it on the scrubbed state)
      active_force(X[0], X[1], C2, fa);
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = clamp_nanprop(fa[i], -1e6f, 1e6f);
      e1 = energy_exact(X);
    }
    const float h1 = clamp_nanprop(e1, -1e10f, 1e10f) + kinetic(P);

    // ---- Metropolis accept (samplers/hmc.py:277-292)
    const float dlt = clamp_nanprop(h0 - h1, -50.0f, 50.0f);
    float acc_p = expf(dlt);
    acc_p = (acc_p > 1.0f) ? 1.0f : acc_p;  // clamp_(max=1); NaN stays NaN and rejects
    const bool accept = init || (L.active && (uu < acc_p));
    if (accept) {
      e_cur = e1;
    } else {  // rejected: bring the parked state and its force back
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const float4 q = *reinterpret_cast<const float4*>(&hmc_smem[park0 + v * (4 * kBlock)]);
        X[2 * v] = v2f{q.x, q.y};
        X[2 * v + 1] = v2f{q.z, q.w};
      }
      const float4 q = *reinterpret_cast<const float4*>(&hmc_smem[fpark]);
      fa[0] = q.x; fa[1] = q.y; fa[2] = q.z; fa[3] = q.w;
    }
    if (init) continue;

    if (a.accept_mask && L.active) a.accept_mask[(int64_t)t * a.n_chains + L.chain] = accept ? 1 : 0;
    if (a.accept_count) {  // wavefront-level count, one atomic per wave
      const unsigned long long b = __ballot(accept && L.active);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(a.accept_count + t, (uint32_t)__popcll(b));
    }
    if (a.traj && --until_keep == 0) {
      until_keep = a.thin;
      store_row(a.traj + traj_row + keep_off, X);
      keep_off += D;
    }
  }
  store_row(a.x + row, X);
}

// Launched IN FRONT of the dense kernel when the energy carries an active-column mask: the kernel whose body does not
// match the mask returns at once (a wave-uniform read of the mask: no host read of device memory).
bool hmc_slot1_applies(const ebm_energy_t& e, const rows::Geometry& geo, int32_t mass_kind, bool diag) {
  return e.kind == EBM_ENERGY_GMM && e.aux != nullptr && e.n_comp >= 1 && e.n_comp <= 8 && geo.G == 1 && geo.NV == 8 &&
         geo.full && mass_kind == EBM_MASS_NONE && !diag;
}

void launch_slot1(dim3 grid, hipStream_t st, HmcArgs a) {
  const size_t smem = ((size_t)kTabFloats + (size_t)(8 + 1) * 4 * kBlock) * sizeof(float);
  hipLaunchKernelGGL((hmc_slot1_kernel<8>), grid, dim3(kBlock), smem, st, a);
}

}  // namespace hmc
}  // namespace ebm
