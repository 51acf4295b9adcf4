// Fused HMC transition kernel for gfx950: momentum draw, Hamiltonian, L safe-mode leapfrog
// steps and the Metropolis accept for n_mh transitions in ONE launch, state in VGPRs.
// Layout and energies: rows.h.  Reference: torchebm/samplers/hmc.py:243-312,
// torchebm/integrators/leapfrog.py:156-185, torchebm/core/base_integrator.py:875-889.
#include <cstdlib>

#include "hmc_kernel.h"

namespace ebm {
using namespace rows;

namespace hmc {
// one definition per energy, each in its own translation unit
void launch_double_well(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
void launch_harmonic(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
void launch_gaussian(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
void launch_gmm(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
// ... and the variants that emit diagnostics records at the kept transitions (hmc_*_diag.hip)
void launch_double_well_diag(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
void launch_harmonic_diag(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
void launch_gaussian_diag(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
void launch_gmm_diag(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
// hmc_literal.hip: the audit form for the element-wise energies (ebm_hmc_chain_audit_f32)
void launch_literal_double_well(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
void launch_literal_harmonic(const rows::Geometry&, dim3, size_t, hipStream_t, const HmcArgs&);
// hmc_ring.hip: the mixture whose means differ in columns 0..3 only, at four waves per SIMD
bool hmc_slot1_applies(const ebm_energy_t&, const rows::Geometry&, int32_t mass_kind);
void launch_slot1(dim3, hipStream_t, HmcArgs);
// hmc_gmm32.hip: any other mixture of up to eight components at dim 32, identity mass, at four waves per SIMD
bool hmc_gmm32_applies(const ebm_energy_t&, const rows::Geometry&, int32_t mass_kind);
void launch_gmm32(dim3, hipStream_t, HmcArgs);
}  // namespace hmc
using hmc::HmcArgs;

bool gauss_hmc_mfma_supported(int32_t dim, int32_t mass_kind);
bool gauss_hmc_shift_supported(int32_t dim);  // gauss_hmc_shift.hip: widths off multiples of 4, 17 .. 158, on shifted rows
int launch_hmc_chain_gauss_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t, double,
                                 const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*, uint64_t, uint64_t,
                                 float*, hipStream_t);
int launch_hmc_chain_gauss_shift_diag(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t,
                                      double, const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*, uint64_t,
                                      uint64_t, float*, hipStream_t);
bool gauss_hmc_stream_supported(const ebm_energy_t& e, int32_t dim);  // gauss_hmc_stream.hip: dims 164 .. 256 with the pre-split image
bool gauss_hmc_stream_shift_supported(const ebm_energy_t& e, int32_t dim);  // gauss_hmc_stream_shift.hip: ... and the widths between them
int launch_hmc_chain_gauss_stream_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t,
                                        double, const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*, uint64_t,
                                        uint64_t, float*, hipStream_t);
int launch_hmc_chain_gauss_stream_shift_diag(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t,
                                             double, const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*,
                                             uint64_t, uint64_t, float*, hipStream_t);
int launch_hmc_chain_gauss_stream(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t, double,
                                  const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*, uint64_t, uint64_t, hipStream_t);
bool gmm_hmc_mfma_supported(int32_t dim, int32_t n_comp, int32_t mass_kind);
bool gmm_hmc_shift_supported(int32_t dim, int32_t n_comp, int32_t mass_kind, bool records);  // gmm_hmc_shift.hip
int launch_hmc_chain_gmm_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t, double,
                               const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*, uint64_t, uint64_t,
                               float*, hipStream_t);
int launch_hmc_chain_gmm_shift_diag(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t,
                                    double, const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*, uint64_t,
                                    uint64_t, float*, hipStream_t);
bool gmm_hmc_wide_supported(int32_t dim, int32_t n_comp, int32_t mass_kind);        // gmm_hmc_wide.hip: mixtures at 132 .. 224 dims
bool gmm_hmc_wide_shift_supported(int32_t dim, int32_t n_comp, int32_t mass_kind);  // gmm_hmc_wide_shift.hip: ... and the widths between
int launch_hmc_chain_gmm_wide(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t, double,
                              const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*, uint64_t, uint64_t, hipStream_t);
int launch_hmc_chain_gmm_wide_shift(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*, int32_t, double,
                                    const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*, uint64_t, uint64_t,
                                    hipStream_t);
bool matrix_hmc_diag_plan(const ebm_energy_t&, int64_t, int32_t, diag::DiagArgs&);
int launch_hmc_chain_matrix_diag(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*,
                                 int32_t, double, const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*,
                                 uint64_t, uint64_t, float*, hipStream_t);
int launch_hmc_chain_gmm_mfma(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*,
                              int32_t, double, const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*,
                              uint64_t, uint64_t, hipStream_t);
int launch_hmc_chain_gauss_mfma(const ebm_energy_t&, float*, int64_t, int32_t, int32_t, int32_t, float, const float*,
                                int32_t, double, const float*, int32_t, float*, uint8_t*, uint32_t*, const float*, const float*,
                                uint64_t, uint64_t, hipStream_t);

// Lane geometry of the transition kernel for this energy / row width (shared by the launcher and the
// diagnostics layout query, which must agree).
static bool hmc_geometry(const ebm_energy_t& e, int32_t dim, Geometry& geo) {
  if (!pick_geometry(dim, geo)) return false;
  // dim-32 full rows can be re-shaped to (G, NV) = (4,2) | (2,4) | (1,8); EBM_HMC_NV overrides
  static const int nv_env = ab_int("EBM_HMC_NV");
  // (not (1, 8) for a mixture of more than eight components: with identity mass that geometry exists only as hmc_ring.hip /
  //  hmc_gmm32.hip, which stop at K = 8 -- launch_geo<GMM> would launch nothing and the call would return success: ADVICE r5)
  if (dim == 32 && (nv_env == 2 || nv_env == 4 || nv_env == 8) && !(nv_env == 8 && e.kind == EBM_ENERGY_GMM && e.n_comp > 8))
    geo = Geometry{8 / nv_env, nv_env, true};
  // measured on MI355X (profiles/r01_bench_kernels.jsonl): the small-mixture energy is fastest with
  // ONE lane per chain -- the means become wave-uniform scalar operands (no LDS traffic, no cross-lane
  // reduction): 1.22 ms per 10 transitions vs 1.76 for (2,4) and 2.3 for the generic (8,1)
  else if (dim == 32 && nv_env == 0 && e.kind == EBM_ENERGY_GMM && e.n_comp <= 8) geo = Geometry{1, 8, true};
  // element-wise energies at dim 32: two lanes x four vectors (one DPP level for E and K): 0.59 vs 0.70 ms
  else if (dim == 32 && nv_env == 0 && (e.kind == EBM_ENERGY_DOUBLE_WELL || e.kind == EBM_ENERGY_HARMONIC))
    geo = Geometry{2, 4, true};
  // ... and at dim 64 / 128 four vectors per lane on 4 / 8 lanes (0.32 vs 0.36 ms at dim 128)
  else if ((dim == 64 || dim == 128) && nv_env == 0 && (e.kind == EBM_ENERGY_DOUBLE_WELL || e.kind == EBM_ENERGY_HARMONIC))
    geo = Geometry{dim / 16, 4, true};
  // ... and THREE vectors per lane where a power-of-two group leaves more than a quarter of its lanes without a vector
  // (rows of 2^k + 1 .. 1.5 2^k vectors: dims 9 .. 12, 17 .. 24, 33 .. 48, 65 .. 96, 129 .. 192, 257 .. 384, 513 .. 768)
  else if (nv_env == 0 && (e.kind == EBM_ENERGY_DOUBLE_WELL || e.kind == EBM_ENERGY_HARMONIC)) {
    const int nvec = (dim + 3) / 4;
    int g3 = 1;
    while (3 * g3 < nvec) g3 <<= 1;
    if (g3 <= 64 && 3 * g3 < geo.G * geo.NV) geo = Geometry{g3, 3, false};
  }
  return true;
}

// Records from the matrix-layout kernels where they run and their layout does not depend on the mass form (the layout
// query is not told it): dense Gaussians and mixtures at dims 20 .. 96.
static bool hmc_matrix_records(const ebm_energy_t& e, int64_t n_chains, int32_t dim, diag::DiagArgs& d) {
  static const bool gauss_rows = ab_switch("EBM_GAUSS_ROWS");
  static const bool gmm_rows = ab_switch("EBM_GMM_ROWS");
  if ((e.kind == EBM_ENERGY_GAUSSIAN && gauss_rows) || (e.kind == EBM_ENERGY_GMM && gmm_rows)) return false;
  // widths off multiples of 4 from 21: shifted rows, the records of their alignment classes interleaved (diag.h)
  static const bool no_shift = ab_switch("EBM_GAUSS_NOSHIFT");
  if (e.kind == EBM_ENERGY_GAUSSIAN && gauss_hmc_shift_supported(dim) && !no_shift) return diag::plan_classes(n_chains, dim, d);
  if (gauss_hmc_stream_shift_supported(e, dim) && !no_shift) return diag::plan_classes(n_chains, dim, d);
  if (e.kind == EBM_ENERGY_GMM && gmm_hmc_shift_supported(dim, e.n_comp, EBM_MASS_NONE, true) && !no_shift) return diag::plan_classes(n_chains, dim, d);
  return matrix_hmc_diag_plan(e, n_chains, dim, d);
}

bool hmc_diag_plan(const ebm_energy_t& e, int64_t n_chains, int32_t dim, diag::DiagArgs& d) {
  if (hmc_matrix_records(e, n_chains, dim, d)) return true;
  Geometry geo;
  if (!hmc_geometry(e, dim, geo)) return false;
  return diag::plan(n_chains, dim, (int64_t)(kBlock / geo.G) * dim, d);
}

int launch_hmc_chain(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh,
                     int32_t n_leapfrog, float eps, const float* eps_table, int32_t mass_kind,
                     double mass_scalar, const float* mass_diag, int32_t thin, float* traj,
                     uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise,
                     const float* u, uint64_t seed, uint64_t offset, float* diag_partials, hipStream_t st) {
  if (diag_partials) {
    diag::DiagArgs dm;
    if (hmc_matrix_records(e, n_chains, dim, dm) && dm.E < 0 && gauss_hmc_stream_shift_supported(e, dim))
      return launch_hmc_chain_gauss_stream_shift_diag(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar,
                                                      mass_diag, thin, traj, accept_mask, accept_count, p_noise, u, seed, offset,
                                                      diag_partials, st);
    if (hmc_matrix_records(e, n_chains, dim, dm) && dm.E < 0 && e.kind == EBM_ENERGY_GMM)
      return launch_hmc_chain_gmm_shift_diag(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag,
                                             thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, diag_partials, st);
    if (hmc_matrix_records(e, n_chains, dim, dm) && dm.E < 0)
      return launch_hmc_chain_gauss_shift_diag(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag,
                                               thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, diag_partials, st);
    if (hmc_matrix_records(e, n_chains, dim, dm))
      return launch_hmc_chain_matrix_diag(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag,
                                          thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, diag_partials, st);
  }
  // (records beyond those shapes: the lane-group kernels)
  if (!diag_partials && e.kind == EBM_ENERGY_GAUSSIAN && gauss_hmc_shift_supported(dim)) {
    // A/B switch: EBM_GAUSS_NOSHIFT=1 keeps the lane-group kernel for widths off multiples of 4
    static const bool no_shift = ab_switch("EBM_GAUSS_NOSHIFT");
    if (!no_shift)
      return launch_hmc_chain_gauss_shift(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag,
                                          thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, nullptr, st);
  }
  if (!diag_partials && e.kind == EBM_ENERGY_GAUSSIAN && gauss_hmc_mfma_supported(dim, mass_kind)) {
    // A/B switch for tests and profiling: EBM_GAUSS_ROWS=1 keeps the LDS mat-vec kernel
    static const bool force_rows = ab_switch("EBM_GAUSS_ROWS");
    if (!force_rows)
      return launch_hmc_chain_gauss_mfma(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar,
                                         mass_diag, thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, st);
  }
  if (!diag_partials && e.kind == EBM_ENERGY_GMM && gmm_hmc_shift_supported(dim, e.n_comp, mass_kind, false)) {
    static const bool no_shift = ab_switch("EBM_GAUSS_NOSHIFT");
    if (!no_shift)
      return launch_hmc_chain_gmm_shift(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag,
                                        thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, nullptr, st);
  }
  if (!diag_partials && gauss_hmc_stream_shift_supported(e, dim)) {
    static const bool no_shift = ab_switch("EBM_GAUSS_NOSHIFT");
    if (!no_shift)
      return launch_hmc_chain_gauss_stream_shift(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag,
                                                 thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, nullptr, st);
  }
  if (!diag_partials && gauss_hmc_stream_supported(e, dim))
    return launch_hmc_chain_gauss_stream(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag, thin,
                                         traj, accept_mask, accept_count, p_noise, u, seed, offset, st);
  // Mixtures of up to 32 components, dims 20 .. 96: the two K x dim passes of the gradient on the bf16 matrix pipe
  // (gauss_hmc_mfma.hip: GmmE).  One shape stays on the lane-group kernel: dim 32 with K <= 8, where one lane per chain
  // with the means as scalar operands (and the active-column body) is faster -- dense means, ms per 10 transitions at
  // L = 20, 2^18 chains (scripts/bench_gmm_dense.py): dim 32: K = 8 0.96 there vs 1.43 here, K = 16 / 32 4.30 / 7.86 vs
  // 1.63 / 2.38; dim 64: K = 8 / 16 / 32 4.20 / 8.71 / 16.0 vs 3.07 / 3.52 / 5.41.
  if (!diag_partials && e.kind == EBM_ENERGY_GMM && gmm_hmc_mfma_supported(dim, e.n_comp, mass_kind) &&
      !(dim == 32 && e.n_comp <= 8)) {
    // A/B switch for tests and profiling: EBM_GMM_ROWS=1 keeps the lane-group kernels
    static const bool force_rows = ab_switch("EBM_GMM_ROWS");
    if (!force_rows)
      return launch_hmc_chain_gmm_mfma(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar,
                                       mass_diag, thin, traj, accept_mask, accept_count, p_noise, u, seed, offset, st);
  }
  // Mixtures at 129 .. 224 dims (no mass vector, no records): five to seven tiles of the same body, where they beat the lane-group kernels
  if (!diag_partials && e.kind == EBM_ENERGY_GMM &&
      (gmm_hmc_wide_supported(dim, e.n_comp, mass_kind) || gmm_hmc_wide_shift_supported(dim, e.n_comp, mass_kind))) {
    static const bool force_rows = ab_switch("EBM_GMM_ROWS");
    if (!force_rows)
      return (gmm_hmc_wide_supported(dim, e.n_comp, mass_kind) ? launch_hmc_chain_gmm_wide : launch_hmc_chain_gmm_wide_shift)(
          e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag, thin, traj, accept_mask, accept_count,
          p_noise, u, seed, offset, st);
  }
  Geometry geo;
  if (!hmc_geometry(e, dim, geo)) return fail(EBM_EDIM, "ebm_hmc_chain_f32: dim %d > 1024 is not supported by the fused kernel", dim);
  HmcArgs a{};
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
  a.park_offset_floats = (int)(smem / sizeof(float));
  // the accepted state and its force (two force slots for NV == 4: hmc_kernel.h F_TWO), one float4 slot per lane and vector
  if (geo.NV >= 4) smem += (size_t)kBlock * geo.NV * 16 * (geo.NV <= 4 ? 3 : 2);
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0};
  a.diag_offset_floats = (int)(smem / sizeof(float));
  const bool diag_kernel = diag_partials != nullptr;
  if (diag_kernel) {
    if (!diag::plan(n_chains, dim, (int64_t)(kBlock / geo.G) * dim, a.diag))
      return fail(EBM_EDIM, "ebm_hmc_chain_f32: diagnostics records are not available for dim %d", dim);
    a.diag.partials = diag_partials;
    // scratch rows, then (narrow rows only: the wide ones reuse the state's parking slot) the tile
    smem += (size_t)(geo.NV >= 4 ? diag::scratch_floats(a.diag.S) : diag::lds_floats(a.diag.E, a.diag.S)) * sizeof(float);
  }
  const int64_t blocks = blocks_for(n_chains, geo);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "ebm_hmc_chain_f32: too many chains for one launch");
  const dim3 grid((unsigned)blocks);
  // A mixture at dim 32 with K <= 8 and identity mass, one lane per chain: two kernels are launched back to back and each
  // reads the active-column mask itself -- hmc_ring.hip does the work when the means differ inside ONE four-column slot
  // (gmm_single_slot: any slot, not only the first) and returns at once otherwise; hmc_gmm32.hip does it for every other
  // mask and returns at once for a single slot.  (Both index chains with 32 bits: a dim-32 state of 2^32 chains is 512 GiB.)
  if ((hmc::hmc_slot1_applies(e, geo, mass_kind) || hmc::hmc_gmm32_applies(e, geo, mass_kind)) && n_chains >= (1LL << 32))
    return fail(EBM_EINVAL, "ebm_hmc_chain_f32: more than 2^32 - 1 chains in one launch");
  if (hmc::hmc_slot1_applies(e, geo, mass_kind)) hmc::launch_slot1(grid, st, a);
  if (hmc::hmc_gmm32_applies(e, geo, mass_kind)) {  // (returns at once when the mask names a single slot)
    hmc::launch_gmm32(grid, st, a);
    return check_launch("ebm_hmc_chain_f32");
  }
  if (diag_kernel) {
    switch (e.kind) {
      case EBM_ENERGY_DOUBLE_WELL: hmc::launch_double_well_diag(geo, grid, smem, st, a); break;
      case EBM_ENERGY_HARMONIC:    hmc::launch_harmonic_diag(geo, grid, smem, st, a); break;
      case EBM_ENERGY_GAUSSIAN:    hmc::launch_gaussian_diag(geo, grid, smem, st, a); break;
      default:                     hmc::launch_gmm_diag(geo, grid, smem, st, a); break;
    }
    return check_launch("ebm_hmc_chain_f32");
  }
  switch (e.kind) {
    case EBM_ENERGY_DOUBLE_WELL: hmc::launch_double_well(geo, grid, smem, st, a); break;
    case EBM_ENERGY_HARMONIC:    hmc::launch_harmonic(geo, grid, smem, st, a); break;
    case EBM_ENERGY_GAUSSIAN:    hmc::launch_gaussian(geo, grid, smem, st, a); break;
    default:                     hmc::launch_gmm(geo, grid, smem, st, a); break;
  }
  return check_launch("ebm_hmc_chain_f32");
}

// ebm_hmc_chain_audit_f32: the literal leapfrog sequence (hmc_kernel.h: leapfrog_literal) -- element-wise energies, dim <= 256
int launch_hmc_chain_audit(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog, float eps,
                           const float* eps_table, int32_t mass_kind, double mass_scalar, const float* mass_diag, int32_t thin,
                           float* traj, uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise, const float* u,
                           uint64_t seed, uint64_t offset, hipStream_t st) {
  const char* who = "ebm_hmc_chain_audit_f32";
  if (e.kind != EBM_ENERGY_DOUBLE_WELL && e.kind != EBM_ENERGY_HARMONIC)
    return fail(EBM_EKIND, "%s: the audit form exists for the element-wise energies (double well, harmonic)", who);
  Geometry geo;
  if (!pick_geometry(dim, geo) || geo.NV != 1) return fail(EBM_EDIM, "%s: dim %d > 256", who, dim);
  geo.full = false;
  HmcArgs a{};
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
  a.park_offset_floats = (int)(smem / sizeof(float));
  a.diag = diag::DiagArgs{nullptr, 0, 0, 0};
  a.diag_offset_floats = (int)(smem / sizeof(float));
  const int64_t blocks = blocks_for(n_chains, geo);
  if (blocks > 0x7fffffffLL) return fail(EBM_EINVAL, "%s: too many chains for one launch", who);
  const dim3 grid((unsigned)blocks);
  if (e.kind == EBM_ENERGY_DOUBLE_WELL) hmc::launch_literal_double_well(geo, grid, smem, st, a);
  else hmc::launch_literal_harmonic(geo, grid, smem, st, a);
  return check_launch(who);
}

}  // namespace ebm
