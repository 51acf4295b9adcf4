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

#include "gauss_mfma_body.h"

namespace ebm {
namespace {

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
// (more than eight components: from 12 dims -- the one-tile kernel costs what it costs at 32, and the lane-group kernels'
//  K x dim passes grow with K: K = 16, 2^16 chains x 20 steps, dims 13 / 16 / 19: 0.152 / 0.157 / 0.252 ms there, 0.103 here)
bool gmm_mfma_supported(int32_t dim, int32_t n_comp) { return dim >= (n_comp > 8 ? 12 : 20) && dim <= 128 && (dim % 4) == 0 && n_comp >= 1 && n_comp <= 32; }

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
bool gauss_res_shift_supported(const ebm_energy_t& e, int32_t dim);  // gauss_res_shift.hip: widths off multiples of 4 up to 254, per-class images
int launch_langevin_chain_gauss_res_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                          const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);
bool gmm_wide_supported(int32_t dim, int32_t n_comp);        // gmm_wide.hip: mixtures at 132 .. 256 dims (five to eight tiles)
bool gmm_wide_shift_supported(int32_t dim, int32_t n_comp);  // gmm_wide_shift.hip: ... and the widths off multiples of 4 between 126 and 254
int launch_langevin_chain_gmm_wide(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                   const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);
int launch_langevin_chain_gmm_wide_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                         const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);
bool gmm_shift_supported(int32_t dim, int32_t n_comp);  // gmm_shift.hip
int launch_langevin_chain_gmm_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                    const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);
bool gauss_shift_supported(int32_t dim);  // gauss_shift.hip
int launch_langevin_chain_gauss_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, float, float, float,
                                      const float*, int, float, float, int32_t, float*, const float*, uint64_t, uint64_t, float*, hipStream_t);

bool matrix_langevin_diag_plan(const ebm_energy_t& e, int64_t n_chains, int32_t dim, diag::DiagArgs& d) {
  // widths off multiples of 4 from 21: shifted rows, the records of their alignment classes interleaved
  static const bool no_shift = ab_switch("EBM_GAUSS_NOSHIFT");
  if (e.kind == EBM_ENERGY_GAUSSIAN && gauss_shift_supported(dim) && !no_shift) return diag::plan_classes(n_chains, dim, d);
  if (gauss_res_shift_supported(e, dim) && !no_shift) return diag::plan_classes(n_chains, dim, d);
  if (e.kind == EBM_ENERGY_GMM && gmm_shift_supported(dim, e.n_comp) && !no_shift) return diag::plan_classes(n_chains, dim, d);
  static const bool no_wide = ab_switch("EBM_GMM_NOWIDE");
  if (e.kind == EBM_ENERGY_GMM && gmm_wide_shift_supported(dim, e.n_comp) && !no_wide) return diag::plan_classes(n_chains, dim, d);
  if (e.kind == EBM_ENERGY_GMM && gmm_wide_supported(dim, e.n_comp) && !no_wide) return diag::plan(n_chains, dim, 32 * (int64_t)dim, d);
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
  if (mixture && (gmm_wide_supported(dim, e.n_comp) || gmm_wide_shift_supported(dim, e.n_comp)))  // five to eight tiles
    return (gmm_wide_supported(dim, e.n_comp) ? launch_langevin_chain_gmm_wide : launch_langevin_chain_gmm_wide_shift)(
        e, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin, cmax, thin, traj, noise, seed, offset,
        diag_partials, st);
  if (a.diag.E < 0 && mixture)  // interleaved classes: the shifted-row kernels
    return launch_langevin_chain_gmm_shift(e, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin, cmax,
                                           thin, traj, noise, seed, offset, diag_partials, st);
  if (a.diag.E < 0 && gauss_res_shift_supported(e, dim))
    return launch_langevin_chain_gauss_res_shift(e, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin, cmax,
                                                 thin, traj, noise, seed, offset, diag_partials, st);
  if (a.diag.E < 0)
    return launch_langevin_chain_gauss_shift(e, x, n_chains, dim, k_steps, eta, sqrt_eta, noise_coef, coef_table, clamp_on, cmin, cmax,
                                             thin, traj, noise, seed, offset, diag_partials, st);
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
