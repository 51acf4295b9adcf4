// The streamed shifted-row Gaussian HMC kernels WITH diagnostics records (gauss_hmc_stream_shift.hip under EBM_SHIFT_DIAG).
#define EBM_SHIFT_DIAG 1
#include "gauss_hmc_stream_shift.hip"
