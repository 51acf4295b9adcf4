// The shifted-row mixture HMC kernels WITH diagnostics records (gmm_hmc_shift.hip under EBM_SHIFT_DIAG: a translation unit
// of its own so that the two halves compile in parallel).
#define EBM_SHIFT_DIAG 1
#include "gmm_hmc_shift.hip"
