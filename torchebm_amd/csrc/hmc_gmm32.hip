// HMC transition kernel for a GENERAL mixture of up to eight isotropic Gaussians at dim 32, identity mass: one lane per
// chain, four waves per SIMD (round 4; the sibling of hmc_ring.hip, which takes the mixtures whose means differ in
// columns 0..3 only).  Replaces the dense body of hmc_kernel.h for this shape (256 VGPRs, two waves per SIMD).
//
// Reference: torchebm/samplers/hmc.py:243-312, torchebm/integrators/leapfrog.py:156-185,
// core/base_integrator.py:875-889; the energy is the build's own.
//
//   * x and p are the only rows that live through a trajectory (64 VGPRs); the weighted mean of the second pass is
//     accumulated in 32 more and becomes the force in place, consumed by the kick on the spot.  The force is NOT carried
//     from transition to transition (32 registers across the momentum draw would cost the fourth wave): a trajectory
//     is L + 1 evaluations, kick (half, whole ..., half) behind each, drift between them;
//   * every lane needs the SAME mu_k[d], so the means are wave-uniform operands: both K x dim passes stream the rows
//     through scalar registers, s_load_dwordx16 double-buffered (volatile asm: left alone the compiler hoists all 2 x 256
//     loads and spills), used directly as the SGPR-pair operand of v_pk_fma_f32.  No LDS traffic, no cross-lane step;
//   * logits in the dot-product form, base 2: l_k = c_k + (log2 e / sigma^2) x . mu_k (softmax is shift invariant: |x|^2
//     drops out), H0 / H1 from the reference's difference form;
//   * safe mode: one running NaN-propagating maximum over the forces and the entry momentum; a chain that leaves the
//     bounds is redone from its parked state by the literal sequence (see hmc_ring.hip), cold code.
#include "hmc_lane.h"

namespace ebm {
namespace hmc {

using namespace lane;

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
constexpr int kTab = 16;                            // LDS head: the eight logit offsets, the eight log-weights
constexpr int kParkFloats = (NV + 1) * 4 * kBlock;  // parked state [v][thread] float4 + one float4 of scalars per lane

template <bool DIAG>
__device__ __forceinline__ void dense_body(const HmcArgs& a) {
  const uint32_t chain32 = blockIdx.x * (uint32_t)kBlock + threadIdx.x;  // n_chains < 2^32 (checked by the launcher, hmc.hip)
  const bool active = (int64_t)chain32 < a.n_chains;
  auto chain_now = [&]() -> uint64_t {  // the only per-lane address register that lives through the kernel (hmc_ring.hip)
    uint32_t c = chain32;
    asm volatile("" : "+v"(c));
    return active ? (uint64_t)c : 0ull;
  };
  // LDS addresses are formed where they are used, from the chain index (a register that holds one through the kernel
  // is spilled to scratch -- even the constant 0 of the table's base)
  auto lane_slot = [&]() -> int {
    uint32_t c = chain32;
    asm volatile("" : "+v"(c));
    return kTab + 4 * (int)(c & (uint32_t)(kBlock - 1));
  };
  auto tab_at = [&](int i) -> const float* {
    int z;
    asm volatile("v_mov_b32 %0, 0" : "=v"(z));
    return hmc_smem + z + i;
  };
  float* const tab = hmc_smem;
  const int K = a.energy.n_comp;
  const float invs2 = a.energy.s1, inv2s2 = a.energy.s0;
  const uint64_t mu_base = (uint64_t)(uintptr_t)a.energy.dev0;

  // chunk c of a pass: 16 floats, row k = c / 2 (padding components re-read the last row), half h = c % 2
  auto issue = [&](int c, v16f& dst) {
    const int k = c >> 1;
    const uint64_t src = mu_base + (uint64_t)(((k < K ? k : K - 1) * D + (c & 1) * 16) * 4);
    asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=s"(dst) : "s"(src));
  };
  auto arrive = [](v16f& v) { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(v)); };

  // per component: the x-independent part of the base-2 logit (lw_k - |mu_k|^2 / (2 sigma^2)) log2 e, and the log-weight:
  // an LDS table read at every evaluation (two broadcast reads; as scalar registers they were spilled to scratch)
  {
    v16f buf[2];
    issue(0, buf[0]);
    float nrm = 0.0f;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      arrive(buf[c & 1]);
      if (c + 1 < 16) issue(c + 1, buf[(c + 1) & 1]);
      const v16f m = buf[c & 1];
#pragma unroll
      for (int i = 0; i < 16; ++i) nrm = __builtin_fmaf(m[i], m[i], nrm);
      if (c & 1) {
        const int k = c >> 1;
        const float lwk = k < K ? a.energy.dev1[k < K ? k : 0] : -__builtin_inff();
        if (threadIdx.x == 0) {
          tab[k] = __builtin_fmaf(-nrm, inv2s2, lwk) * kLog2e;
          tab[8 + k] = lwk;
        }
        nrm = 0.0f;
      }
    }
    __syncthreads();
  }
  const float s_logit = invs2 * kLog2e;

  // F = -dE/dx = (sum_k r_k mu_k - x) / sigma^2,  r = softmax_k(c_k + x . mu_k / sigma^2)
  auto force = [&](const v2f (&X)[NP], v2f (&F)[NP]) {
    float logit[8];
    v16f buf[2];
    issue(0, buf[0]);
    const float* const tb = tab_at(0);
    const float4 c2a = *reinterpret_cast<const float4*>(tb), c2b = *reinterpret_cast<const float4*>(tb + 4);
    const float c2[8] = {c2a.x, c2a.y, c2a.z, c2a.w, c2b.x, c2b.y, c2b.z, c2b.w};
    v2f da, db;  // two chains: a packed FMA cannot feed the next one back to back
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      arrive(buf[c & 1]);
      if (c + 1 < 16) issue(c + 1, buf[(c + 1) & 1]);
      const v16f m = buf[c & 1];
      const int h = c & 1;
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        if (h == 0 && q == 0) {
          da = X[0] * v2f{m[0], m[1]};
          db = X[1] * v2f{m[2], m[3]};
        } else {
          da = pk_fma(X[8 * h + q], v2f{m[2 * q], m[2 * q + 1]}, da);
          db = pk_fma(X[8 * h + q + 1], v2f{m[2 * q + 2], m[2 * q + 3]}, db);
        }
      }
      if (h == 1) {
        da += db;
        logit[c >> 1] = __builtin_fmaf(da.x + da.y, s_logit, c2[c >> 1]);
      }
      asm volatile("" : "+v"(da), "+v"(db));  // the chunk's FMAs stay between its load and the next one
      __builtin_amdgcn_sched_barrier(0);
    }
    float top = __builtin_fmaxf(__builtin_fmaxf(logit[0], logit[1]), logit[2]);  // a NaN logit resurfaces in the sum
    top = __builtin_fmaxf(__builtin_fmaxf(top, logit[3]), logit[4]);
    top = __builtin_fmaxf(__builtin_fmaxf(top, logit[5]), logit[6]);
    top = __builtin_fmaxf(top, logit[7]);
    float w[8], sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      w[k] = __builtin_amdgcn_exp2f(logit[k] - top);
      sum += w[k];
    }
    issue(0, buf[0]);
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      arrive(buf[c & 1]);
      if (c + 1 < 16) issue(c + 1, buf[(c + 1) & 1]);
      const v16f m = buf[c & 1];
      const int h = c & 1;
      const v2f wk = splat(w[c >> 1]);
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        if (c < 2) F[8 * h + q] = wk * v2f{m[2 * q], m[2 * q + 1]};
        else F[8 * h + q] = pk_fma(wk, v2f{m[2 * q], m[2 * q + 1]}, F[8 * h + q]);
        asm volatile("" : "+v"(F[8 * h + q]));  // keep the FMAs with their chunk
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    const v2f sc = splat(__builtin_amdgcn_rcpf(sum) * invs2), ni = splat(-invs2);
#pragma unroll
    for (int j = 0; j < NP; ++j) F[j] = pk_fma(F[j], sc, X[j] * ni);
  };

  // E(x) = -logsumexp_k(lw_k - |x - mu_k|^2 / (2 sigma^2)), the reference's difference form (H0 / H1, records)
  auto energy_exact = [&](const v2f (&X)[NP]) -> float {
    float logit[8];
    v16f buf[2];
    issue(0, buf[0]);
    const float* const tb = tab_at(8);
    const float4 lwa = *reinterpret_cast<const float4*>(tb), lwb = *reinterpret_cast<const float4*>(tb + 4);
    const float lw[8] = {lwa.x, lwa.y, lwa.z, lwa.w, lwb.x, lwb.y, lwb.z, lwb.w};
    v2f da, db;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
      arrive(buf[c & 1]);
      if (c + 1 < 16) issue(c + 1, buf[(c + 1) & 1]);
      const v16f m = buf[c & 1];
      const int h = c & 1;
#pragma unroll
      for (int q = 0; q < 8; q += 2) {
        const v2f ea = X[8 * h + q] - v2f{m[2 * q], m[2 * q + 1]};
        const v2f eb = X[8 * h + q + 1] - v2f{m[2 * q + 2], m[2 * q + 3]};
        if (h == 0 && q == 0) {
          da = ea * ea;
          db = eb * eb;
        } else {
          da = pk_fma(ea, ea, da);
          db = pk_fma(eb, eb, db);
        }
      }
      if (h == 1) {
        da += db;
        logit[c >> 1] = __builtin_fmaf(-(da.x + da.y), inv2s2, lw[c >> 1]);
      }
      asm volatile("" : "+v"(da), "+v"(db));
      __builtin_amdgcn_sched_barrier(0);
    }
    float top = logit[0];
#pragma unroll
    for (int k = 1; k < 8; ++k) top = __builtin_fmaxf(top, logit[k]);
    top = __builtin_fabsf(top) == __builtin_inff() ? 0.0f : top;  // torch.logsumexp's rule for an infinite maximum
    float sum = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; ++k) sum += __expf(logit[k] - top);
    return -(top + logf(sum));
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

  auto load_row = [&](const float* __restrict__ src, v2f (&R)[NP]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      float4 q = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
      if (active) q = *reinterpret_cast<const float4*>(src + 4 * v);
      R[2 * v] = v2f{q.x, q.y};
      R[2 * v + 1] = v2f{q.z, q.w};
    }
  };
  auto store_row = [&](float* __restrict__ dst, const v2f (&R)[NP]) {
    if (active) {
#pragma unroll
      for (int v = 0; v < NV; ++v)
        *reinterpret_cast<float4*>(dst + 4 * v) = make_float4(R[2 * v].x, R[2 * v].y, R[2 * v + 1].x, R[2 * v + 1].y);
    }
  };
  auto unpark = [&](v2f (&R)[NP]) {
    const int park0 = lane_slot();
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      const float4 q = *reinterpret_cast<const float4*>(&hmc_smem[park0 + v * (4 * kBlock)]);
      R[2 * v] = v2f{q.x, q.y};
      R[2 * v + 1] = v2f{q.z, q.w};
    }
  };
  v2f X[NP];
  load_row(a.x + chain_now() * D, X);
  auto draw_momentum = [&](int t, v2f (&P)[NP]) {
    if (a.p_noise) {
      load_row(a.p_noise + ((uint64_t)t * (uint64_t)a.n_chains + chain_now()) * D, P);
    } else {
      const uint64_t step = a.step0 + 2ull * (uint64_t)t;
      uint32_t c = chain32;
      asm volatile("" : "+v"(c));  // (see hmc_ring.hip: the counters are not to be hoisted out of the transition loop)
      const uint64_t g0 = (uint64_t)c * (uint64_t)(D / 4);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        const F4 n = normal4_at(a.key, g0 + (uint64_t)v, step);
        P[2 * v] = v2f{n.v[0], n.v[1]};
        P[2 * v + 1] = v2f{n.v[2], n.v[3]};
        if (v & 1) __builtin_amdgcn_sched_barrier(0);
      }
    }
  };

  int until_keep = a.thin;
  int keep = 0;
  float eps = a.eps;
  float e_cur = energy_exact(X);

  for (int t = 0; t < a.n_mh; ++t) {
    if (a.eps_table) eps = to_sgpr(a.eps_table[t]);  // wave-uniform: a scalar register
    const float half_eps = 0.5f * eps;
    const int n_lf = a.n_leapfrog;

    v2f P[NP];
    draw_momentum(t, P);
    float uu;
    if (a.u) uu = active ? a.u[(uint64_t)t * (uint64_t)a.n_chains + chain_now()] : 2.0f;
    else {
      uint32_t c = chain32;
      asm volatile("" : "+v"(c));
      uu = u01_half_open(pick(philox_at(a.key, (uint64_t)(c >> 2), a.step0 + 2ull * (uint64_t)t + 1ull), (int)(c & 3)));
    }
    const float h0 = clamp_nanprop(e_cur, -1e10f, 1e10f) + kinetic(P);

    // park the accepted state and the scalars that are only needed again behind the trajectory
    {
      const int park0 = lane_slot();
#pragma unroll
      for (int v = 0; v < NV; ++v)
        *reinterpret_cast<float4*>(&hmc_smem[park0 + v * (4 * kBlock)]) = make_float4(X[2 * v].x, X[2 * v].y, X[2 * v + 1].x, X[2 * v + 1].y);
      *reinterpret_cast<float4*>(&hmc_smem[park0 + NV * (4 * kBlock)]) = make_float4(e_cur, 0.0f, h0, uu);
    }

    // ---- the trajectory, common path: L + 1 evaluations, the kick (half, whole, ..., whole, half) behind each, the drift
    //      between them.  m: running NaN-propagating maximum of |force| and of the entry momentum / 1e24
    float m = 0.0f;
#pragma unroll
    for (int j = 0; j < NP; ++j) m = max3np(m, __builtin_fabsf(P[j].x) * 1e-24f, __builtin_fabsf(P[j].y) * 1e-24f);  // |p| >= 1e30 <=> > 1e6
    asm volatile("" : "+v"(m));  // here, not sunk behind the trajectory (that would keep the drawn momentum alive)
    const v2f E2 = splat(eps);
    for (int l = 0; l <= n_lf; ++l) {
      v2f F[NP];
      force(X, F);
      const v2f K2 = splat((l == 0 || l == n_lf) ? half_eps : eps);
#pragma unroll
      for (int j = 0; j < NP; ++j) {
        m = max3np(m, __builtin_fabsf(F[j].x), __builtin_fabsf(F[j].y));
        P[j] = pk_fma(K2, F[j], P[j]);
      }
      if (l < n_lf) {
#pragma unroll
        for (int j = 0; j < NP; ++j) X[j] = pk_fma(E2, P[j], X[j]);
      }
    }
    float e1 = energy_exact(X);
    const bool bad = !(m <= 1e6f) || !(__builtin_fabsf(e1) < __builtin_inff());
    if (__builtin_expect(bad, 0)) {
      // ---- the literal sequence from the parked state (leapfrog.py:165-185):  per step  f = clamp(-dE/dx(x)); p += eps/2 f;
      //      x += eps p; f' = clamp(-dE/dx(x)); p += eps/2 f'; scrub x, p.  -dE/dx as autograd returns it: NaN in every
      //      coordinate when every squared distance overflows (E = +inf, or NaN)
      unpark(X);
      draw_momentum(t, P);
      for (int l = 0; l < n_lf; ++l) {
        for (int h = 0; h < 2; ++h) {  // (a loop: ONE inlined copy of the kick)
          v2f F[NP];
          force(X, F);
          const bool blown = !(energy_exact(X) < __builtin_inff());
          const float nan = __builtin_nanf("");
#pragma unroll
          for (int j = 0; j < NP; ++j) {
            P[j].x = __builtin_fmaf(half_eps, blown ? nan : clamp_nanprop(F[j].x, -1e6f, 1e6f), P[j].x);
            P[j].y = __builtin_fmaf(half_eps, blown ? nan : clamp_nanprop(F[j].y, -1e6f, 1e6f), P[j].y);
          }
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
      e1 = energy_exact(X);
    }
    const float h1 = clamp_nanprop(e1, -1e10f, 1e10f) + kinetic(P);
    const float4 parked = *reinterpret_cast<const float4*>(&hmc_smem[lane_slot() + NV * (4 * kBlock)]);

    // ---- Metropolis accept (samplers/hmc.py:277-292)
    const float dlt = clamp_nanprop(parked.z - h1, -50.0f, 50.0f);
    float acc_p = expf(dlt);
    acc_p = (acc_p > 1.0f) ? 1.0f : acc_p;  // clamp_(max=1); NaN stays NaN and rejects
    const bool accept = active && (parked.w < acc_p);
    if (accept) {
      e_cur = e1;
    } else {  // rejected: bring the parked state back
      e_cur = parked.x;
      unpark(X);
    }

    if (a.accept_mask && active) a.accept_mask[(uint64_t)t * (uint64_t)a.n_chains + chain_now()] = accept ? 1 : 0;
    if (a.accept_count) {  // wavefront-level count, one atomic per wave
      const unsigned long long b = __ballot(accept && active);
      if ((threadIdx.x & 63) == 0 && b) atomicAdd(a.accept_count + t, (uint32_t)__popcll(b));
    }
    if ((a.traj != nullptr || DIAG) && --until_keep == 0) {
      until_keep = a.thin;
      if (a.traj) store_row(a.traj + (chain_now() * (uint64_t)a.n_kept + (uint64_t)keep) * D, X);
      if constexpr (DIAG) {
        // samplers/hmc.py:294-310; the tile is the parking area of the state (dead until the next transition parks again)
        float* const tile = hmc_smem + kTab;
        float* const scratch = tile + kParkFloats;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < NV; ++v)
          *reinterpret_cast<float4*>(tile + (int)threadIdx.x * D + 4 * v) = make_float4(X[2 * v].x, X[2 * v].y, X[2 * v + 1].x, X[2 * v + 1].y);
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

template <bool DIAG>
__global__ __launch_bounds__(kBlock, 4) void hmc_gmm32_kernel(HmcArgs a) {
  // the mixtures whose means differ in columns 0..3 only have their own kernels (hmc_ring.hip), launched in front
  if (gmm_single_slot(a.energy) >= 0) return;
  dense_body<DIAG>(a);
}

bool hmc_gmm32_applies(const ebm_energy_t& e, const rows::Geometry& geo, int32_t mass_kind) {
  return e.kind == EBM_ENERGY_GMM && e.n_comp >= 1 && e.n_comp <= 8 && geo.G == 1 && geo.NV == 8 && geo.full &&
         mass_kind == EBM_MASS_NONE;
}

void launch_gmm32(dim3 grid, hipStream_t st, HmcArgs a) {
  size_t smem = (size_t)(kTab + kParkFloats) * sizeof(float);
  if (a.diag.partials) {
    smem += (size_t)diag::scratch_floats(a.diag.S) * sizeof(float);
    hipLaunchKernelGGL(hmc_gmm32_kernel<true>, grid, dim3(kBlock), smem, st, a);
  } else {
    hipLaunchKernelGGL(hmc_gmm32_kernel<false>, grid, dim3(kBlock), smem, st, a);
  }
}

}  // namespace hmc
}  // namespace ebm
