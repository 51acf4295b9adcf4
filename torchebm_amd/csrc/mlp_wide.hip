// k-fused Langevin chain (and stand-alone energy / gradient) for the two-hidden-layer SiLU MLP energy
//     E(x) = w3 . silu(W2 silu(W1 x + b1) + b2) + b3,      x in R^D,  D <= 128,  hidden width H in {64, 128, 256}
// -- the reference's benchmark MLP (benchmarks/registry.py:372-387: Linear(dim, 128) ... at dim 8 / 32 / 128) beyond the
// 2-D two-moons network that mlp.hip specialises (there the first layer is two FMAs per hidden unit on the VALU).
//
// All FOUR contractions of one evaluation run on the exact-f32 matrix cores (v_mfma_f32_32x32x2_f32), and every
// operand that is not a weight is used where the previous contraction left it:
//   a1^T[i, m] = sum_c W1[i, c] x[c, m]        h1 = silu(a1 + b1)       s1 = silu'(a1 + b1)
//   a2^T[j, m] = sum_i W2[j, i] h1[i, m]       E += w3[j] silu(a2 + b2) d2 = w3[j] silu'(a2 + b2)
//   T^T[i, m]  = sum_j W2[j, i] d2[j, m]       d1 = T * s1
//   g^T[c, m]  = sum_i W1[i, c] d1[i, m]
// A wavefront owns 32 chains: lane l = (m, h), m = l & 31 the chain, h = l >> 5 the K-half of the MFMA.  Everything is
// kept TRANSPOSED (rows = feature index, columns = chain) in the 32x32 C/D layout -- register r of lane (m, h) holds
// row (r & 3) + 8 (r >> 2) + 4 h of a 32-row tile -- INCLUDING THE STATE x: the K index of every contraction is
// enumerated in the order that layout holds it (K-step (tile, r): half h contributes k = 32 tile + row_of(r, h)), so
// the B operand of each MFMA is simply register r of the previous result, the gradient comes out in the layout the
// state lives in, and no value ever changes lanes.  The A operands are single LDS words: W1 with row stride
// Dpad + 1 and W2 with row stride H + 1 are conflict-free for both the row walk (forward) and the column walk
// (backward).  Registers: x, s1, d2, T and g are 16 H/32 (or 16 Dpad/32) values each -- up to 320 at D = H = 128 --
// so the kernel runs one wave per SIMD on the unified 512-entry register file (accumulators in AGPRs); the LDS
// (W1 + W2 = 132 KiB at D = H = 128) allows one workgroup per CU anyway.
// H = 256 (STREAM): W2 alone is 256 KiB, so no weight is staged -- the A operands come straight from the row-major
// parameter block in global memory (0.3 MB, L2-resident; every workgroup reads the same words): the forward walks
// (K contiguous in a weight row) take one 16-byte load per four K-steps, the transposed walks one coalesced dword per
// K-step (32 consecutive words of a weight row per K-half), requested two to sixteen stages ahead of the MFMAs that
// consume them.  The LDS instead parks s1 = silu'(a1) (128 values per lane, idle across both W2 contractions), which
// keeps the live set at two hidden-width tiles sets + the state: ~330-400 of the 512 registers.
// Reference: torchebm/samplers/langevin_dynamics.py:154-185 (the loop), core/base_integrator.py:711-731 (the update).
#include "mlp_wide_body.h"

namespace ebm {
using namespace widemlp;

int launch_mlp_stream(const widemlp::WideArgs& a, int dt, hipStream_t st, const char* who);  // mlp_stream.hip

bool mlp_wide_supported(int32_t hidden, int32_t dim) {
#ifdef EBM_MLP_H256  // (make H256=1: the streamed-weight family of round 2, mlp_stream*.hip -- not in the shipped library since round 5:
  return (hidden == 64 || hidden == 128 || hidden == 256) && dim >= 1 && dim <= 128;  // the reference's network is 128 wide)
#else
  return (hidden == 64 || hidden == 128) && dim >= 1 && dim <= 128;
#endif
}

// k_steps == 0: evaluation into energy_out / grad_out; else the k-fused chain
int launch_mlp_wide(int32_t hidden, const float* params, float* x, int64_t n_chains, int32_t dim, int32_t k_steps, float eta,
                    float sqrt_eta, float noise_coef, const float* coef_table, int clamp_on, float cmin, float cmax,
                    int32_t thin, float* traj, const float* noise, uint64_t seed, uint64_t offset, float* energy_out,
                    float* grad_out, float* diag_partials, const void* w1_image, hipStream_t st, const char* who,
                    const uint64_t* rng_dev) {
  WideArgs a{};
  a.rng_dev = rng_dev;
  a.w1_image = static_cast<const char*>(w1_image);
  a.x = x; a.n_chains = n_chains; a.dim = dim; a.k_steps = k_steps;
  a.eta = eta; a.sqrt_eta = sqrt_eta; a.noise_coef = noise_coef;
  a.table = reinterpret_cast<const float4*>(coef_table);
  a.clamp_on = clamp_on; a.cmin = cmin; a.cmax = cmax;
  a.thin = thin; a.n_kept = thin > 0 ? k_steps / thin : 0; a.traj = traj; a.noise = noise;
  a.key = RngKey{(uint32_t)seed, (uint32_t)(seed >> 32)};
  a.step0 = offset; a.params = params; a.energy_out = energy_out; a.grad_out = grad_out;
  a.diag_partials = diag_partials; a.diag_blocks = ceil_div64(n_chains, 32);
  const int dt = (dim + 31) / 32;
#define EBM_WIDE(HTV)                                    \
  switch (dt) {                                          \
    case 1: return launch_one<HTV, 1>(a, st, who);       \
    case 2: return launch_one<HTV, 2>(a, st, who);       \
    case 3: return launch_one<HTV, 3>(a, st, who);       \
    default: return launch_one<HTV, 4>(a, st, who);      \
  }
  if (hidden == 64) { EBM_WIDE(2) }
#ifdef EBM_MLP_H256
  if (hidden == 256) return launch_mlp_stream(a, dt, st, who);
#endif
  EBM_WIDE(4)
#undef EBM_WIDE
}

}  // namespace ebm

#ifdef EBM_PHASE_TIMES
// scripts/mlp_phase_times.py: the shader-clock log of wave 0 (mlp_wide_body.h, EBM_STAMP)
extern "C" __attribute__((visibility("default"))) int ebm_debug_phase_log(unsigned long long* out, int n) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(ebm::widemlp::ebm_phase_log), (size_t)n * sizeof(unsigned long long));
}
#endif
