// The shifted-row Gaussian HMC kernels WITH diagnostics records (gauss_hmc_shift.hip under EBM_SHIFT_DIAG: a translation
// unit of its own so that the two halves compile in parallel).
#define EBM_SHIFT_DIAG 1
#include "gauss_hmc_shift.hip"
