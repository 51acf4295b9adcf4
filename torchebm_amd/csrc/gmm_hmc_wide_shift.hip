// The 129 .. 256-dim mixture HMC kernels on SHIFTED rows (gmm_hmc_wide.hip under EBM_WIDE_SH: widths off multiples of 4).
#define EBM_WIDE_SH 1
#include "gmm_hmc_wide.hip"
