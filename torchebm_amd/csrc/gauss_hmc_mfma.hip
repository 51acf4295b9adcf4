// HMC transitions for the dense Gaussian energy on the matrix cores: launchers (the body: mfma_hmc_body.h).
#include "mfma_hmc_body.h"

namespace ebm {
namespace {

// dims 32 / 64 run best held to 256 VGPRs (two waves per SIMD: 0.62 vs 0.80 ms at dim 64) on the exact-f32 contraction,
// dims 96 / 128 need more than that for the state alone; the bf16x3 form of two tiles needs the transient split
// registers and runs better unconstrained (0.53 vs 0.97 ms, dim 64).  (Several entry points because hipcc 7.2 silently
// ignores a template-dependent __launch_bounds__ argument.)
template <int NT, bool DIAGM, bool B3>
int launch_nt_b(const GaussHmcArgs& a, hipStream_t st) {
  // the last 16 coordinates of the last tile all padding: that K-block is left out of every contraction (bf16 form)
  if constexpr (B3 && NT >= 2)
    if (32 * NT - a.dim >= 16) return launch_policy<NT, DIAGM, GaussE<NT, true, 1>, 0>(a, st);
  return launch_policy<NT, DIAGM, GaussE<NT, B3>, (NT == 1 || (NT == 2 && !B3)) ? 2 : 0>(a, st);
}

template <int NT, bool DIAGM>
int launch_nt(const GaussHmcArgs& a, hipStream_t st) {
  // A/B switch for tests and profiling: EBM_GAUSS_F32MFMA=1 keeps the exact-f32 MFMA contraction
  static const bool f32_mfma = ab_switch("EBM_GAUSS_F32MFMA");
  return f32_mfma ? launch_nt_b<NT, DIAGM, false>(a, st) : launch_nt_b<NT, DIAGM, true>(a, st);
}

}  // namespace

bool gauss_hmc_mfma_supported(int32_t dim, int32_t mass_kind) {
  // measured against the lane-group kernel (profiles/r02_bench_gauss_hmc_{mfma,rows}.jsonl, ms per 10 transitions, L = 10,
  // 2^16 chains): dim 32: 0.20 vs 0.41, dim 64: 0.45 vs 1.32, dim 96: 1.19 vs 3.50, dim 100: 2.27 vs 3.89, dim 128: 2.26 vs 11.5
  (void)mass_kind;
  // round 3: five tiles (dims 132 .. 160) -- 150 KB of split operands still fit the CU's LDS; beyond, the sampler's GEMM route
  return dim >= 20 && dim <= 160 && (dim % 4) == 0;
}

int launch_hmc_chain_gauss_mfma(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh,
                                int32_t n_leapfrog, float eps, const float* eps_table, int32_t mass_kind,
                                double mass_scalar, const float* mass_diag, int32_t thin, float* traj, uint8_t* accept_mask,
                                uint32_t* accept_count, const float* p_noise, const float* u, uint64_t seed,
                                uint64_t offset, hipStream_t st) {
  const GaussHmcArgs a = matrix_hmc_args(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag,
                                         thin, traj, accept_mask, accept_count, p_noise, u, seed, offset);
  if (a.mass_diag) {
    switch ((dim + 31) / 32) {
      case 1: return launch_nt<1, true>(a, st);
      case 2: return launch_nt<2, true>(a, st);
      case 3: return launch_nt<3, true>(a, st);
      case 4: return launch_nt<4, true>(a, st);
      default: return launch_nt<5, true>(a, st);
    }
  }
  switch ((dim + 31) / 32) {
    case 1: return launch_nt<1, false>(a, st);
    case 2: return launch_nt<2, false>(a, st);
    case 3: return launch_nt<3, false>(a, st);
    case 4: return launch_nt<4, false>(a, st);
    default: return launch_nt<5, false>(a, st);
  }
}

}  // namespace ebm
