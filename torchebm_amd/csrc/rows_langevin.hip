// k-fused Langevin chain for energies whose gradient couples the coordinates of a chain
// (Gaussian, Gaussian mixture), and the stand-alone energy / gradient kernel.
// Layout and energies: rows.h.  Reference: torchebm/samplers/langevin_dynamics.py:154-185,
// torchebm/core/base_integrator.py:711-731, torchebm/core/base_model.py:181-210.
#include "rows.h"

namespace ebm {
using namespace rows;

namespace {

extern __shared__ __attribute__((aligned(16))) float rows_smem[];

struct RowChainArgs {
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
  EnergyParams energy;
  int param_floats;
  diag::DiagArgs diag;     // per-block diagnostics records at the kept steps (null: off)
  int diag_offset_floats;  // start of the diagnostics tile in dynamic LDS
};

template <int KIND, int G, int NV, bool FULL, bool HEUN>
__device__ __forceinline__ void langevin_chain_rows_body(const RowChainArgs& a) {
  using LaneT = Lane<G, NV, FULL>;
  LaneT L;
  L.init(a.n_chains, a.dim);
  const Smem S = carve_smem<NV>(rows_smem, a.param_floats);
  stage_params(a.energy, a.dim, S.param);
  Energy<KIND, LaneT> en;
  en.init(a.energy, L, S);

  const int64_t row = L.active ? L.chain * (int64_t)a.dim : 0;
  Slice<NV> x;
  load_slice(L, a.x, row, x);
  const int64_t traj_row = L.active ? L.chain * (int64_t)a.n_kept * a.dim : 0;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  int keep = 0;
  const bool keeping = a.traj != nullptr || a.diag.partials != nullptr;
  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;

  for (int s = 0; s < a.k_steps; ++s) {
    if (a.table) {
      const float4 t = a.table[s];
      eta = t.x; sqrt_eta = t.y; noise_coef = t.z;
    }
    Slice<NV> g, eps;
    en.template eval<false>(L, x, g);
    if constexpr (HEUN) {  // predictor x1 = x - eta*g0, corrector gradient 0.5*g0 + 0.5*g(x1)
      Slice<NV> x1, g1;
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) x1.a[v][i] = L.ok(v, i) ? x.a[v][i] - eta * g.a[v][i] : 0.0f;
      en.template eval<false>(L, x1, g1);
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int i = 0; i < 4; ++i) g.a[v][i] = 0.5f * g.a[v][i] + 0.5f * g1.a[v][i];
    }
    if (a.noise) load_slice(L, a.noise, ((int64_t)s * a.n_chains) * a.dim + row, eps);
    else normal_slice(L, a.key, a.step0 + (uint64_t)s, eps);
#pragma unroll
    for (int v = 0; v < NV; ++v)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // reference op order (base_integrator.py:397,728-729): rounded mul, rounded add, ...
        const float x1 = x.a[v][i] - eta * g.a[v][i];
        const float dw = eps.a[v][i] * sqrt_eta;
        float nv = x1 + noise_coef * dw;
        if (a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
        x.a[v][i] = L.ok(v, i) ? nv : 0.0f;
      }
    if (keeping && --until_keep == 0) {
      until_keep = a.thin;
      if (a.traj) {
        store_slice(L, a.traj, traj_row + keep_off, x);
        keep_off += a.dim;
      }
      if (a.diag.partials) {  // langevin_dynamics.py:170-185: mean / var of the population, mean energy
        float* tile = rows_smem + a.diag_offset_floats;
        tile_store(L, tile, x);
        Slice<NV> g_unused;
        const float e_now = en.template eval<true>(L, x, g_unused);
        diag::emit(a.diag, keep, tile, tile + a.diag.E, tile_valid<G>(a.n_chains, a.dim), a.dim, (L.active && L.lg == 0) ? e_now : 0.0f, 0.0f);
        ++keep;
      }
    }
  }
  store_slice(L, a.x, row, x);
}

template <int KIND, int G, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void langevin_chain_rows_kernel(RowChainArgs a) {
  if constexpr (KIND == EBM_ENERGY_GMM && G == 1 && FULL && NV >= 4) {  // means that differ in columns 0..3 only: rows.h kGmmSlot1
    if (gmm_is_slot1(a.energy)) {
      langevin_chain_rows_body<kGmmSlot1, G, NV, FULL, false>(a);
      return;
    }
  }
  langevin_chain_rows_body<KIND, G, NV, FULL, false>(a);
}

template <int KIND, int G, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void langevin_heun_rows_kernel(RowChainArgs a) {
  langevin_chain_rows_body<KIND, G, NV, FULL, true>(a);
}

// ---------------------------------------------------------------------------------
// dim == 2 (the 2-D Gaussian of BASELINE config 1, 2-D mixtures): TWO chains per lane.
// A Philox counter covers four consecutive flat elements = two whole rows, so with one chain per lane every lane
// pays the ten rounds and both Box-Muller pairs for two normals it uses and two it discards -- and the RNG is
// 85 % of a 2-D step.  Here the lane's float4 is the pair of rows (2p, 2p+1): one counter, four normals, all used;
// the energy body is the one-lane-per-chain body of rows.h evaluated twice (same arithmetic, bit for bit, as the
// one-chain-per-lane kernel this replaces), the update is the same expression on four slots.
// ---------------------------------------------------------------------------------
template <int KIND, bool HEUN>
__device__ __forceinline__ void langevin_chain_pair_body(const RowChainArgs& a) {
  using LaneT = Lane<1, 1, false>;
  const int64_t pair = (int64_t)blockIdx.x * kBlock + threadIdx.x;
  LaneT lane[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    LaneT& L = lane[c];
    L.chain = 2 * pair + c;
    L.lg = 0;
    L.chain_in_wave = threadIdx.x & 63;
    L.active = L.chain < a.n_chains;
    L.vec_ok = false;
    L.dim = 2;
    L.col[0] = 0;
    L.valid = L.active ? 0x3u : 0u;
  }
  const Smem S = carve_smem<1>(rows_smem, a.param_floats);
  stage_params(a.energy, a.dim, S.param);
  Energy<KIND, LaneT> en;
  en.init(a.energy, lane[0], S);

  float x[4] = {0.0f, 0.0f, 0.0f, 0.0f};
  const bool vec = lane[1].active && (((uintptr_t)a.x & 15u) == 0);
  if (vec) {
    const float4 t = *reinterpret_cast<const float4*>(a.x + 4 * pair);
    x[0] = t.x; x[1] = t.y; x[2] = t.z; x[3] = t.w;
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane[i >> 1].active) x[i] = a.x[4 * pair + i];
  }
  auto grad = [&](const float (&xs)[4], float (&g)[4]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      Slice<1> xc, gc;
      xc.a[0][0] = xs[2 * c]; xc.a[0][1] = xs[2 * c + 1]; xc.a[0][2] = 0.0f; xc.a[0][3] = 0.0f;
      en.template eval<false>(lane[c], xc, gc);
      g[2 * c] = gc.a[0][0]; g[2 * c + 1] = gc.a[0][1];
    }
  };
  int until_keep = a.thin;
  int keep = 0;
  const bool keeping = a.traj != nullptr || a.diag.partials != nullptr;
  float eta = a.eta, sqrt_eta = a.sqrt_eta, noise_coef = a.noise_coef;
  for (int s = 0; s < a.k_steps; ++s) {
    if (a.table) {
      const float4 t = a.table[s];
      eta = t.x; sqrt_eta = t.y; noise_coef = t.z;
    }
    float g[4];
    grad(x, g);
    if constexpr (HEUN) {
      float x1[4], g1[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) x1[i] = lane[i >> 1].active ? x[i] - eta * g[i] : 0.0f;
      grad(x1, g1);
#pragma unroll
      for (int i = 0; i < 4; ++i) g[i] = 0.5f * g[i] + 0.5f * g1[i];
    }
    F4 eps;
    if (a.noise) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
        eps.v[i] = lane[i >> 1].active ? a.noise[((int64_t)s * a.n_chains) * 2 + 4 * pair + i] : 0.0f;
    } else {
      eps = normal4_at(a.key, (uint64_t)pair, a.step0 + (uint64_t)s);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float x1 = x[i] - eta * g[i];
      const float dw = eps.v[i] * sqrt_eta;
      float nv = x1 + noise_coef * dw;
      if (a.clamp_on) nv = clamp_nanprop(nv, a.cmin, a.cmax);
      x[i] = lane[i >> 1].active ? nv : 0.0f;
    }
    if (keeping && --until_keep == 0) {
      until_keep = a.thin;
      if (a.traj) {
#pragma unroll
        for (int c = 0; c < 2; ++c)
          if (lane[c].active) {
            float* dst = a.traj + (lane[c].chain * (int64_t)a.n_kept + keep) * 2;
            dst[0] = x[2 * c]; dst[1] = x[2 * c + 1];
          }
      }
      if (a.diag.partials) {
        float* tile = rows_smem + a.diag_offset_floats;
        float e_part = 0.0f;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          Slice<1> xc, g_unused;
          xc.a[0][0] = x[2 * c]; xc.a[0][1] = x[2 * c + 1]; xc.a[0][2] = 0.0f; xc.a[0][3] = 0.0f;
          const float e_now = en.template eval<true>(lane[c], xc, g_unused);
          if (lane[c].active) {
            e_part += e_now;
            tile[4 * threadIdx.x + 2 * c] = x[2 * c];
            tile[4 * threadIdx.x + 2 * c + 1] = x[2 * c + 1];
          }
        }
        const int64_t left = a.n_chains - (int64_t)blockIdx.x * (2 * kBlock);
        const int valid = left >= 2 * kBlock ? 4 * kBlock : (left > 0 ? (int)left * 2 : 0);
        diag::emit(a.diag, keep, tile, tile + a.diag.E, valid, 2, e_part, 0.0f);
      }
      ++keep;
    }
  }
  if (vec) {
    *reinterpret_cast<float4*>(a.x + 4 * pair) = make_float4(x[0], x[1], x[2], x[3]);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (lane[i >> 1].active) a.x[4 * pair + i] = x[i];
  }
}

template <int KIND, bool HEUN>
__global__ __launch_bounds__(kBlock) void langevin_chain_pair_kernel(RowChainArgs a) {
  langevin_chain_pair_body<KIND, HEUN>(a);
}

// ---------------------------------------------------------------------------------
// noise-free descent (gradient descent / Nesterov), k fused steps, any analytic energy
// ---------------------------------------------------------------------------------
struct DescentArgs {
  float* x;
  int64_t n_chains;
  int32_t dim;
  int32_t k_steps;
  float eta;
  const float* eta_table;
  int32_t nesterov;
  float mu;
  int32_t thin, n_kept;
  float* traj;
  EnergyParams energy;
  int param_floats;
};

template <int KIND, int G, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void descent_chain_rows_kernel(DescentArgs a) {
  using LaneT = Lane<G, NV, FULL>;
  LaneT L;
  L.init(a.n_chains, a.dim);
  const Smem S = carve_smem<NV>(rows_smem, a.param_floats);
  stage_params(a.energy, a.dim, S.param);
  Energy<KIND, LaneT> en;
  en.init(a.energy, L, S);
  const int64_t row = L.active ? L.chain * (int64_t)a.dim : 0;
  Slice<NV> x, v;
  load_slice(L, a.x, row, x);
#pragma unroll
  for (int q = 0; q < NV; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i) v.a[q][i] = 0.0f;
  const int64_t traj_row = L.active ? L.chain * (int64_t)a.n_kept * a.dim : 0;
  int until_keep = a.thin;
  int64_t keep_off = 0;
  float eta = a.eta;
  for (int s = 0; s < a.k_steps; ++s) {
    if (a.eta_table) eta = a.eta_table[s];
    Slice<NV> g;
    if (a.nesterov) {
      Slice<NV> la;
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) la.a[q][i] = L.ok(q, i) ? __builtin_fmaf(a.mu, v.a[q][i], x.a[q][i]) : 0.0f;
      en.template eval<false>(L, la, g);
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float vn = __builtin_fmaf(-eta, g.a[q][i], v.a[q][i] * a.mu);
          v.a[q][i] = L.ok(q, i) ? vn : 0.0f;
          x.a[q][i] = L.ok(q, i) ? x.a[q][i] + vn : 0.0f;
        }
    } else {
      en.template eval<false>(L, x, g);
#pragma unroll
      for (int q = 0; q < NV; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) x.a[q][i] = L.ok(q, i) ? __builtin_fmaf(-eta, g.a[q][i], x.a[q][i]) : 0.0f;
    }
    if (a.traj && --until_keep == 0) {
      until_keep = a.thin;
      store_slice(L, a.traj, traj_row + keep_off, x);
      keep_off += a.dim;
    }
  }
  store_slice(L, a.x, row, x);
}

struct EgArgs {
  const float* x;
  int64_t n_chains;
  int32_t dim;
  float* e_out;
  float* g_out;
  EnergyParams energy;
  int param_floats;
};

template <int KIND, int G, int NV, bool FULL>
__global__ __launch_bounds__(kBlock) void energy_grad_kernel(EgArgs a) {
  using LaneT = Lane<G, NV, FULL>;
  LaneT L;
  L.init(a.n_chains, a.dim);
  const Smem S = carve_smem<NV>(rows_smem, a.param_floats);
  stage_params(a.energy, a.dim, S.param);
  Energy<KIND, LaneT> en;
  en.init(a.energy, L, S);
  const int64_t row = L.active ? L.chain * (int64_t)a.dim : 0;
  Slice<NV> x, g;
  load_slice(L, a.x, row, x);
  const float e = en.template eval<true>(L, x, g);
  if (a.e_out && L.active && L.lg == 0) a.e_out[L.chain] = e;
  if (a.g_out) store_slice(L, a.g_out, row, g);
}

// Element-wise energies on rows wider than the lane-group geometries take (dim > 1024): one wave per chain,
// lanes stride over the row.  Only the diagnostics / stand-alone evaluation of very wide element-wise
// states come here; the chain kernels for those energies are flat and have no row limit.
template <int KIND>
__global__ __launch_bounds__(kBlock) void energy_grad_wide_row_kernel(const float* __restrict__ x, int64_t n_chains,
                                                                      int32_t dim, float s0, float s1,
                                                                      float* __restrict__ e_out, float* __restrict__ g_out) {
  const int64_t chain = (int64_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
  if (chain >= n_chains) return;
  const int lane = threadIdx.x & 63;
  const float* row = x + chain * (int64_t)dim;
  float acc = 0.0f;
  for (int d = lane; d < dim; d += 64) {
    const float xv = row[d];
    if constexpr (KIND == EBM_ENERGY_DOUBLE_WELL) {
      const float u = xv * xv - s1;
      acc += u * u;
      if (g_out) g_out[chain * (int64_t)dim + d] = ((4.0f * s0) * u) * xv;
    } else {
      acc += xv * xv;
      if (g_out) g_out[chain * (int64_t)dim + d] = (2.0f * s0) * xv;
    }
  }
  acc = group_sum<64>(acc);
  if (e_out && lane == 0) e_out[chain] = s0 * acc;
}

}  // namespace

// Lane geometry of the row-coupled Langevin chain for this energy / row width (shared by the launcher and the
// diagnostics layout query, which must agree).
static bool rows_langevin_pair(const ebm_energy_t& e, int32_t dim) {
  // A/B switch for tests and profiling: EBM_NO_PAIR=1 keeps one chain per lane at dim 2
  static const bool off = ab_switch("EBM_NO_PAIR");
  return !off && dim == 2 && (e.kind == EBM_ENERGY_GAUSSIAN || e.kind == EBM_ENERGY_GMM);
}

static bool rows_langevin_geometry(const ebm_energy_t& e, int32_t dim, int heun, Geometry& geo, bool& lane_per_chain) {
  if (!pick_geometry(dim, geo)) return false;
  // small mixture, dim 16 / 32: one lane per chain, the means become wave-uniform scalar operands
  // (rows.h: small_scalar_mu) -- no LDS traffic and no cross-lane reduction in the step loop
  lane_per_chain = !heun && e.kind == EBM_ENERGY_GMM && e.n_comp <= 8 && (dim == 16 || dim == 32);
  if (lane_per_chain) geo = Geometry{1, dim / 4, true};
  return true;
}

bool rows_langevin_diag_plan(const ebm_energy_t& e, int heun, int64_t n_chains, int32_t dim, diag::DiagArgs& d) {
  Geometry geo;
  bool lpc;
  if (!rows_langevin_geometry(e, dim, heun, geo, lpc)) return false;
  if (heun && (e.kind == EBM_ENERGY_DOUBLE_WELL || e.kind == EBM_ENERGY_HARMONIC)) return false;  // Heun element-wise: flat kernel only
  if (rows_langevin_pair(e, dim)) return diag::plan(n_chains, dim, 4 * (int64_t)kBlock, d);  // two chains per lane
  return diag::plan(n_chains, dim, (int64_t)(kBlock / geo.G) * dim, d);
}

int launch_langevin_chain_rows(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim,
                               int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                               const float* coef_table, int clamp_on, float cmin, float cmax,
                               int32_t thin, float* traj, const float* noise, uint64_t seed,
                               uint64_t offset, int heun, float* diag_partials, hipStream_t st) {
  Geometry geo;
  bool lane_per_chain;
  if (!rows_langevin_geometry(e, dim, heun, geo, lane_per_chain))
    return fail(EBM_EDIM, "ebm_langevin_chain_f32: dim %d > 1024 is not supported for this energy", dim);
  RowChainArgs a{};
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset;
  size_t smem = 0;
  plan_params(e, dim, geo, a.energy, a.param_floats, smem);
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0};
  a.diag_offset_floats = (int)(smem / sizeof(float));
  if (diag_partials) {
    if (!rows_langevin_diag_plan(e, heun, n_chains, dim, a.diag))
      return fail(EBM_EDIM, "ebm_langevin_chain_f32: diagnostics records are not available for this energy / dim %d", dim);
    a.diag.partials = diag_partials;
    smem += (size_t)diag::lds_floats(a.diag.E, a.diag.S) * sizeof(float);
  }
  const bool pair = rows_langevin_pair(e, dim);
  const int64_t blocks = pair ? ceil_div64(n_chains, 2 * (int64_t)kBlock) : blocks_for(n_chains, geo);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_langevin_chain_f32: too many chains for one launch");
  const dim3 grid((unsigned)blocks), block(kBlock);
  if (pair && e.kind == EBM_ENERGY_GAUSSIAN) {
    if (heun) hipLaunchKernelGGL((langevin_chain_pair_kernel<EBM_ENERGY_GAUSSIAN, true>), grid, block, smem, st, a);
    else hipLaunchKernelGGL((langevin_chain_pair_kernel<EBM_ENERGY_GAUSSIAN, false>), grid, block, smem, st, a);
  } else if (pair) {
    if (heun) hipLaunchKernelGGL((langevin_chain_pair_kernel<EBM_ENERGY_GMM, true>), grid, block, smem, st, a);
    else hipLaunchKernelGGL((langevin_chain_pair_kernel<EBM_ENERGY_GMM, false>), grid, block, smem, st, a);
  } else if (heun && e.kind == EBM_ENERGY_GAUSSIAN)
    EBM_GEO_LAUNCH(langevin_heun_rows_kernel, EBM_ENERGY_GAUSSIAN, geo, grid, block, smem, st, a);
  else if (heun && e.kind == EBM_ENERGY_GMM)
    EBM_GEO_LAUNCH(langevin_heun_rows_kernel, EBM_ENERGY_GMM, geo, grid, block, smem, st, a);
  else if (heun)
    return fail(EBM_EKIND, "ebm_langevin_heun_chain_f32: element-wise energies run on the flat kernel");
  else if (lane_per_chain && dim == 32)
    hipLaunchKernelGGL((langevin_chain_rows_kernel<EBM_ENERGY_GMM, 1, 8, true>), grid, block, smem, st, a);
  else if (lane_per_chain)
    hipLaunchKernelGGL((langevin_chain_rows_kernel<EBM_ENERGY_GMM, 1, 4, true>), grid, block, smem, st, a);
  else
    EBM_KIND_LAUNCH(langevin_chain_rows_kernel, e.kind, geo, grid, block, smem, st, a);
  return check_launch(heun ? "ebm_langevin_heun_chain_f32" : "ebm_langevin_chain_f32");
}

int launch_descent_chain(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t k_steps,
                         float eta, const float* eta_table, int32_t nesterov, float momentum, int32_t thin,
                         float* traj, hipStream_t st) {
  Geometry geo;
  if (!pick_geometry(dim, geo)) return fail(EBM_EDIM, "ebm_descent_chain_f32: dim %d > 1024 is not supported", dim);
  geo.full = false;  // deterministic optimiser, not a throughput path: one masked variant per geometry
  DescentArgs a{};
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps; a.eta = eta; a.eta_table = eta_table;
  a.nesterov = nesterov; a.mu = momentum; a.thin = thin; a.n_kept = k_steps / thin; a.traj = traj;
  size_t smem = 0;
  plan_params(e, dim, geo, a.energy, a.param_floats, smem);
  const int64_t blocks = blocks_for(n_chains, geo);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_descent_chain_f32: too many chains for one launch");
  EBM_KIND_LAUNCH(descent_chain_rows_kernel, e.kind, geo, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_descent_chain_f32");
}

int launch_energy_grad(const ebm_energy_t& e, const float* x, int64_t n_chains, int32_t dim,
                       float* e_out, float* g_out, hipStream_t st) {
  Geometry geo;
  if (!pick_geometry(dim, geo)) {
    if (e.kind != EBM_ENERGY_DOUBLE_WELL && e.kind != EBM_ENERGY_HARMONIC)
      return fail(EBM_EDIM, "ebm_energy_grad_f32: dim %d > 1024 is not supported for this energy", dim);
    const int64_t wide_blocks = ceil_div64(n_chains, kWavesPerBlock);
    if (wide_blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_energy_grad_f32: too many chains for one launch");
    if (e.kind == EBM_ENERGY_DOUBLE_WELL)
      hipLaunchKernelGGL(energy_grad_wide_row_kernel<EBM_ENERGY_DOUBLE_WELL>, dim3((unsigned)wide_blocks), dim3(kBlock), 0, st,
                         x, n_chains, dim, e.s[0], e.s[1], e_out, g_out);
    else
      hipLaunchKernelGGL(energy_grad_wide_row_kernel<EBM_ENERGY_HARMONIC>, dim3((unsigned)wide_blocks), dim3(kBlock), 0, st,
                         x, n_chains, dim, e.s[0], e.s[1], e_out, g_out);
    return check_launch("ebm_energy_grad_f32");
  }
  geo.full = false;  // one evaluation per launch: the masked form is as fast, and halves the variants
  EgArgs a{};
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.e_out = e_out; a.g_out = g_out;
  size_t smem = 0;
  plan_params(e, dim, geo, a.energy, a.param_floats, smem);
  const int64_t blocks = blocks_for(n_chains, geo);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_energy_grad_f32: too many chains for one launch");
  EBM_KIND_LAUNCH(energy_grad_kernel, e.kind, geo, dim3((unsigned)blocks), dim3(kBlock), smem, st, a);
  return check_launch("ebm_energy_grad_f32");
}

}  // namespace ebm
