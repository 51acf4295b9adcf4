// HMC transitions for dense Gaussians at dims 164 .. 256: launchers of the plain call (the evaluation: gauss_stream_e.h; the
// transition body: mfma_hmc_body.h; with records: matrix_hmc_diag.hip).
#include "gauss_stream_e.h"

namespace ebm {
namespace {

template <int NT, bool DIAGM>
int launch_stream(const GaussHmcArgs& a, hipStream_t st) {
  return launch_policy<NT, DIAGM, GaussStreamE<NT>, 0>(a, st);
}

}  // namespace

bool gauss_hmc_stream_supported(const ebm_energy_t& e, int32_t dim) {
  return e.kind == EBM_ENERGY_GAUSSIAN && dim > 160 && dim <= 256 && (dim % 4) == 0 && e.aux != nullptr &&
         (reinterpret_cast<uintptr_t>(e.aux) & 15) == 0;
}

int launch_hmc_chain_gauss_stream(const ebm_energy_t& e, float* x, int64_t n_chains, int32_t dim, int32_t n_mh, int32_t n_leapfrog,
                                  float eps, const float* eps_table, int32_t mass_kind, double mass_scalar, const float* mass_diag,
                                  int32_t thin, float* traj, uint8_t* accept_mask, uint32_t* accept_count, const float* p_noise,
                                  const float* u, uint64_t seed, uint64_t offset, hipStream_t st) {
  const GaussHmcArgs a = matrix_hmc_args(e, x, n_chains, dim, n_mh, n_leapfrog, eps, eps_table, mass_kind, mass_scalar, mass_diag,
                                         thin, traj, accept_mask, accept_count, p_noise, u, seed, offset);
  const int nt = (dim + 31) / 32;
  if (a.mass_diag) return nt == 6 ? launch_stream<6, true>(a, st) : (nt == 7 ? launch_stream<7, true>(a, st) : launch_stream<8, true>(a, st));
  return nt == 6 ? launch_stream<6, false>(a, st) : (nt == 7 ? launch_stream<7, false>(a, st) : launch_stream<8, false>(a, st));
}

}  // namespace ebm

#ifdef EBM_PHASE_TIMES
extern "C" __attribute__((visibility("default"))) int ebm_debug_hmc_phase_log(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ebm::ebm_hmc_phase_log), (size_t)n * sizeof(unsigned long long));
}
#endif
