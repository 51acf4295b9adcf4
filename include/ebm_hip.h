/*
 * ebm_hip.h -- C ABI of libebm_hip.so: the MI355X (gfx950) kernels behind torchebm's
 * Langevin / HMC sampler inner loop.
 *
 * The reference (soran-ghaderi/torchebm) is a pure-Python library with no FFI layer;
 * its boundary for this path is the Python class API.  These entry
 * points are what a ctypes binding inside the reference would call; each one names the
 * reference code it replaces (paths relative to the reference checkout).
 *
 * Conventions (all entry points)
 *   - the caller owns every buffer; the library never allocates or frees device memory
 *     and keeps no pointer after a call returns;
 *   - device pointers are fp32, row-major contiguous, 16-byte aligned;
 *   - `stream` is a hipStream_t (NULL = the default stream); calls only ENQUEUE work
 *     and return, they never synchronise the device or the host;
 *   - return value: 0 = ok, >0 = a hipError_t from the launch, <0 = EBM_E* argument
 *     error; the text is available from ebm_last_error_string() (thread local);
 *   - no C++ exception crosses the boundary; no mutable global state; the library reads no
 *     environment variable (the kernel-selection switches of scripts/ -- EBM_GAUSS_ROWS and
 *     friends -- exist only in builds made with -DEBM_AB_SWITCHES; tests/test_abi.py checks
 *     that the shipped object does not import getenv).
 *
 * Random numbers ("native RNG", used when a noise pointer is NULL)
 *   Philox4x32-10 keyed by `seed`.  For step s (64-bit, = offset + local step index)
 *   and flat element e:  counter = { lo32(e/4), hi32(e/4), lo32(s), hi32(s) }, the
 *   four 32-bit outputs o0..o3 give, by Box-Muller on (o0,o1) and (o2,o3), the four
 *   standard normals of elements 4*(e/4)+0..3.  Uniforms (HMC accept) use
 *   u = (o[e%4] >> 8) * 2^-24 in [0,1).  The stream therefore depends only on
 *   (seed, offset, step, element) -- never on launch geometry -- and the fused k-step
 *   kernels, the per-step kernel and ebm_noise_fill_f32 all draw the same field.
 */
#ifndef EBM_HIP_H
#define EBM_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define EBM_ABI_VERSION 8

#if defined(__GNUC__)
#define EBM_API __attribute__((visibility("default")))
#else
#define EBM_API
#endif

/* argument errors (negative); positive return values are hipError_t */
#define EBM_EINVAL  (-1)  /* bad size / NULL pointer / misaligned pointer          */
#define EBM_EKIND   (-2)  /* unknown or unsupported energy kind for this entry     */
#define EBM_EDIM    (-3)  /* dim outside the range the fused kernel supports       */

/* Analytic energies the kernels fuse (reference: torchebm/core/base_model.py). */
enum {
  EBM_ENERGY_DOUBLE_WELL = 0, /* E = h * sum_j (x_j^2 - b^2)^2   base_model.py:130-148  s[0]=h  s[1]=(float)(b*b)           */
  EBM_ENERGY_HARMONIC    = 1, /* E = (0.5*k) * sum_j x_j^2       base_model.py:213-229  s[0]=(float)(0.5*k)                 */
  EBM_ENERGY_GAUSSIAN    = 2, /* E = 0.5 d^T P d, d = x - mu     base_model.py:151-210  dev0=mu[dim] dev1=P[dim*dim] (P=cov^-1)
                                 aux = NULL, or (since ABI version 4) the device image that ebm_gauss_prec_image_f32 built from
                                 THIS dev1 -- multiples of 4 from 132 to 512, and (ABI version 5) the widths 158 .. 254 that
                                 are NOT a multiple of 4, where the image holds one shifted copy per alignment class (4 for an
                                 odd width, 2 for width = 2 mod 4; ebm_gauss_prec_image_bytes gives the size, 0 = no image at
                                 this width): the tiled kernels then move their slabs of P by LDS-direct loads instead of
                                 loading, splitting and storing fp32 rows every stage, and the streamed kernels of the shifted
                                 widths (Langevin 158 .. 254, HMC 161 .. 254) REQUIRE it (without it those widths take the
                                 lane-group kernels).  A stale image gives the old matrix's samples; NULL is always safe.     */
  EBM_ENERGY_GMM         = 3, /* E = -logsumexp_k(logw_k - |x-mu_k|^2 * s[0])  (not in the reference: SURVEY §8 a6)
                                 s[0]=1/(2 sigma^2)  s[1]=1/sigma^2  n_comp=K  dev0=mu[K*dim] dev1=logw[K]
                                 aux = NULL, or device int32[1]: bit v set <=> the component means differ somewhere in
                                 columns 4v..4v+3 (v < 8).  A hint, read by the kernel itself: columns all components
                                 share drop out of the responsibilities, and when the mask is 1 (a mixture of a plane
                                 embedded in the first coordinates, e.g. BASELINE config 3's ring) the one-lane-per-chain
                                 kernels (dim 16 / 32, K <= 8) run their K x dim passes over four columns only.
                                 A wrong mask gives wrong samples; NULL is always safe.                                    */
  EBM_ENERGY_MLP         = 4  /* E = w3 . silu(W2 silu(W1 x + b1) + b2) + b3   (the energy of the reference's
                                 examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-31)
                                 n_comp = hidden width H (64 or 128; 256 only in a build made with `make H256=1`), dim <= 128 (the reference's benchmark network
                                 benchmarks/registry.py:372-387 at dim 8 / 32 / 128),
                                 dev0 = packed fp32 parameters W1[H,dim] b1[H] W2[H,H] b2[H] w3[H] b3[1] (torch Linear layout).
                                 Supported by ebm_langevin_chain_f32, ebm_energy_grad_f32 and ebm_hmc_chain_f32 (EBM_EDIM for
                                 other widths).  (H = 256, where built, reads
                                 the weights from dev0 throughout the launch: dev0 must then be 16-byte aligned.)
                                 aux = NULL, or (H = 128, 64 < dim <= 128, since ABI version 4) the device image that
                                 ebm_mlp_w1_image_f32 built from THIS dev0 (cast to const int32_t*): with it the Langevin
                                 and energy / gradient entries run their contractions on the bf16 matrix pipe with
                                 three-way split operands (fp32 accuracy) as the narrower shapes always do; without it on
                                 the exact-f32 matrix instruction (1.6x slower).  A stale image gives the old network's
                                 samples; NULL is always safe.                                                            */
};

typedef struct ebm_energy {
  int32_t kind;        /* EBM_ENERGY_*                              */
  int32_t n_comp;      /* mixture components (GMM), else 0          */
  float   s[4];        /* scalar parameters, see the enum           */
  const float* dev0;   /* device parameter arrays, see the enum     */
  const float* dev1;
  const int32_t* aux;  /* optional device hints, see the enum (NULL: none) */
} ebm_energy_t;

/* noise kinds for ebm_noise_fill_f32 */
enum { EBM_NOISE_NORMAL = 0, EBM_NOISE_UNIFORM = 1, EBM_NOISE_RAW_U32 = 2 };

/* mass kinds for the HMC / leapfrog entries (reference: samplers/hmc.py:92-159, integrators/leapfrog.py:91-101) */
enum { EBM_MASS_NONE = 0, EBM_MASS_SCALAR = 1, EBM_MASS_DIAG = 2 };

EBM_API int ebm_version(void);
EBM_API const char* ebm_last_error_string(void);

/*
 * One Euler-Maruyama step with an externally supplied gradient.
 * Replaces BaseSDERungeKuttaIntegrator.step with the EulerMaruyama tableau
 * (core/base_integrator.py:673-731, integrators/euler_maruyama.py:55-65) and is the HIP
 * counterpart of the Triton POC kernel (cuda/fused_langevin.py:34-62).
 * Arithmetic, each op rounded to fp32, no FMA contraction:
 *     x1  = x - eta * grad            (= x + eta*(1.0*drift), drift = -grad)
 *     dw  = eps * sqrt_eta            (sqrt_eta  = (float) sqrt((double)eta))
 *     out = x1 + noise_coef * dw      (noise_coef = (float) sqrt(2*sigma^2), 0 => ODE step, no noise drawn)
 *     out = clamp(out, cmin, cmax)    if clamp_on   (samplers/langevin_dynamics.py:166-167)
 * eps = noise[e] if noise != NULL, else the native RNG field at step `offset`.
 * `out` may alias `x`.  `grad` may be NULL (zero drift).
 */
EBM_API int ebm_langevin_step_f32(const float* x, const float* grad, float* out, const float* noise,
                          int64_t n_elem, float eta, float sqrt_eta, float noise_coef,
                          int32_t clamp_on, float cmin, float cmax,
                          uint64_t seed, uint64_t offset, void* stream);

/*
 * The same step with a TENSOR diffusion coefficient D instead of the scalar noise scale
 * (BaseSDERungeKuttaIntegrator.step(..., diffusion=D), core/base_integrator.py:652-671, 724-729):
 *     out = (x - eta * grad) + sqrt(2 * D_e) * (eps * sqrt_eta),      D_e = diffusion[e % diffusion_period]
 * with (2.0 * D) ** 0.5 evaluated as torch does (fp32 product, correctly rounded fp32 square root).
 * diffusion_period = 1 (a 0-dim tensor), the row width (one value per coordinate) or n_elem (a full field); it must
 * divide n_elem.  No clamp (the integrator has none).  `out` may alias `x`, `grad` may be NULL.
 */
EBM_API int ebm_langevin_step_diffusion_f32(const float* x, const float* grad, float* out, const float* noise,
                                            const float* diffusion, int64_t diffusion_period, int64_t n_elem, float eta,
                                            float sqrt_eta, uint64_t seed, uint64_t offset, void* stream);

/*
 * The same step with the RNG coordinates read from DEVICE memory: rng_state = {seed, step} (two
 * uint64).  Kernel arguments are frozen when a launch is captured into a HIP graph; keeping
 * (seed, step) in a device buffer that the graph itself advances lets one captured
 * "gradient + update" iteration be replayed k times per sample() call with fresh noise every time
 * (the launch-bound autograd route of BASELINE config 5).  No injected-noise form.
 */
EBM_API int ebm_langevin_step_dev_f32(const float* x, const float* grad, float* out, int64_t n_elem,
                                      float eta, float sqrt_eta, float noise_coef,
                                      int32_t clamp_on, float cmin, float cmax,
                                      const uint64_t* rng_state, void* stream);

/*
 * k fused Langevin steps for an analytic energy: gradient + EM update + noise + clamp
 * + thinned trajectory stores, state resident in registers/LDS across the k steps.
 * Replaces the hot loop of LangevinDynamics.sample (samplers/langevin_dynamics.py:154-185)
 * including BaseModel.gradient (core/base_model.py:62-127) for the fused energies.
 *   x            [n_chains, dim]    in/out
 *   coef_table   NULL, or device float[k_steps][4] = {eta_i, sqrt_eta_i, noise_coef_i, 0}
 *                (pre-expanded schedulers, core/schedulable.py:55-75); when NULL the three
 *                scalars are used for every step
 *   traj         NULL, or [n_chains, k_steps/thin, dim]; row j is the state after step (j+1)*thin
 *   diag_partials NULL, or the per-block diagnostics records of the k_steps/thin kept steps (see
 *                ebm_diag_layout / ebm_diag_finish_f32 below): the sampler diagnostics of
 *                samplers/langevin_dynamics.py:170-185 (population mean / var, mean energy) without leaving
 *                the launch.  EBM_EDIM / EBM_EKIND when this energy / dim has no in-kernel form (ask
 *                ebm_diag_layout first).
 *   noise        NULL (native RNG, steps offset .. offset+k-1), or [k_steps, n_chains, dim]
 *   clamp_on     a flag word (ABI 8; 0 / 1 mean what they always did): EBM_CHAIN_CLAMP = 1 clamps to [cmin, cmax];
 *                EBM_CHAIN_CONTRACTED = 2 PERMITS contracted arithmetic -- x^2 - b^2 and x - eta g as fused multiply-adds, the
 *                coefficient noise_coef * sqrt_eta folded into the Box-Muller radius -- where a kernel has that form (element-wise
 *                energies, the plain call: constant coefficients, no clamp, no trajectory, no records, the kernels' own draws:
 *                8 of 76 vector instructions per float4 group-step fewer); every other call ignores the bit.  Not the
 *                reference's rounding (core/base_integrator.py:711-731 rounds every multiply and add): same law, same Philox
 *                field, last-bit differences per step.  Host side: `sampler.fused_arithmetic = True` on LangevinDynamics.
 */
#define EBM_CHAIN_CLAMP      1
#define EBM_CHAIN_CONTRACTED 2
EBM_API int ebm_langevin_chain_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                           int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                           const float* coef_table, int32_t clamp_on, float cmin, float cmax,
                           int32_t thin, float* traj, float* diag_partials, const float* noise,
                           uint64_t seed, uint64_t offset, void* stream);

/*
 * The same k fused steps with the reference's Heun (improved Euler) drift update,
 * LangevinDynamics(integrator="heun"): tableau a = ((), (1,)), b = (1/2, 1/2) (integrators/heun.py)
 * evaluated in the op order of BaseSDERungeKuttaIntegrator (core/base_integrator.py:387-397, 711-731):
 *   k0 = -grad E(x);  x1 = x + eta*(1*k0);  k1 = -grad E(x1);
 *   x' = x + eta*(0.5*k0 + 0.5*k1) + noise_coef*(eps*sqrt_eta)
 * Two gradient evaluations per step, one noise draw (same Philox coordinates as the EM chain).
 * Arguments as ebm_langevin_chain_f32; DoubleWell / Harmonic / Gaussian / mixture energies
 * (EBM_ENERGY_MLP returns EBM_EKIND).
 */
EBM_API int ebm_langevin_heun_chain_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                           int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                           const float* coef_table, int32_t clamp_on, float cmin, float cmax,
                           int32_t thin, float* traj, float* diag_partials, const float* noise,
                           uint64_t seed, uint64_t offset, void* stream);

/*
 * n_mh fused HMC transitions for an analytic energy (or EBM_ENERGY_MLP): momentum draw, Hamiltonian,
 * L leapfrog steps (safe mode: force clamp +-1e6, NaN scrub), Metropolis accept.
 * Replaces the hot loop of HamiltonianMonteCarlo.sample (samplers/hmc.py:243-312),
 * LeapfrogIntegrator.integrate (integrators/leapfrog.py:116-187) and the clamps of
 * core/base_integrator.py:875-889.
 *   x            [n_chains, dim]   in/out
 *   eps_table    NULL, or device float[n_mh] (scheduled step size per MH step)
 *   mass         EBM_MASS_*: none / scalar (a double, as the Python float it mirrors: the kernel uses
 *                (float)sqrt(m) for the momentum draw, (float)m for the kinetic energy and
 *                (float)max(m,1e-10) in the drift) / device float[dim]
 *   traj         NULL, or [n_chains, n_mh/thin, dim]
 *   diag_partials NULL, or the per-block diagnostics records of the n_mh/thin kept transitions (samplers/hmc.py:294-310:
 *                population mean / var, mean clamped energy, acceptance rate), see ebm_diag_layout below
 *   accept_mask  NULL, or uint8[n_mh, n_chains]   (1 = proposal accepted)
 *   accept_count NULL, or uint32[n_mh], must be zeroed by the caller; receives the number
 *                of accepted chains per MH step (wavefront ballot + one atomic per wave)
 *   p_noise      NULL (native RNG), or [n_mh, n_chains, dim] standard normals
 *   u            NULL (native RNG), or [n_mh, n_chains] uniforms in [0,1)
 * Native RNG consumes two steps per transition: offset+2t (momentum), offset+2t+1 (uniform).
 */
EBM_API int ebm_hmc_chain_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                      int32_t n_mh, int32_t n_leapfrog, float eps, const float* eps_table,
                      int32_t mass_kind, double mass_scalar, const float* mass_diag,
                      int32_t thin, float* traj, float* diag_partials, uint8_t* accept_mask, uint32_t* accept_count,
                      const float* p_noise, const float* u,
                      uint64_t seed, uint64_t offset, void* stream);

/*
 * Leapfrog sub-steps with an externally supplied force (opaque drift closure):
 * LeapfrogIntegrator.step / .integrate (integrators/leapfrog.py:63-187).
 *   kick_drift:  f = clamp(force) if safe;  p_half = p + (0.5*eps)*f;  x_new = x + eps*p_half [/ max(m,1e-10)]
 *   kick:        f = clamp(force) if safe;  p_new  = p_half + (0.5*eps)*f;  if safe: nan_to_num(x_new, p_new)
 * `x_new` of kick is updated in place (the NaN scrub); outputs may alias inputs.
 */
EBM_API int ebm_leapfrog_kick_drift_f32(const float* x, const float* p, const float* force,
                                float* x_new, float* p_half, int64_t n_chains, int32_t dim,
                                float eps, int32_t mass_kind, double mass_scalar,
                                const float* mass_diag, int32_t safe, void* stream);
EBM_API int ebm_leapfrog_kick_f32(float* x_new, const float* p_half, const float* force, float* p_new,
                          int64_t n_elem, float eps, int32_t safe, void* stream);

/*
 * Metropolis accept for the per-step HMC path (samplers/hmc.py:277-292):
 *   d = clamp(h0 - h1, -50, 50); a = min(1, exp(d)); acc = u < a; x = acc ? x_prop : x
 * u = u_or_null[c], or the native uniform field at step `offset`.
 */
EBM_API int ebm_hmc_accept_f32(float* x, const float* x_prop, const float* h0, const float* h1,
                       const float* u, uint8_t* accept_mask, uint32_t* accept_count,
                       int64_t n_chains, int32_t dim, uint64_t seed, uint64_t offset, void* stream);

/*
 * ABI 6 -- the AUDIT form of ebm_hmc_chain_f32 for the element-wise energies (double well, harmonic; dim <= 256; no records):
 * same arguments, same transitions, but the trajectory is the reference's safe-mode leapfrog step LITERALLY
 * (torchebm/integrators/leapfrog.py:156-185): the force re-evaluated at the top of every step, two separate half kicks, every
 * multiply and add rounded on its own, the drift divided by max(m, 1e-10) per step, both nan_to_num_ scrubs on every step,
 * energy and force re-evaluated at the top of every transition (samplers/hmc.py:243-256).  Given the same accept decisions the
 * state is the reference's BIT FOR BIT (tests/test_hmc_audit_gpu.py: torch.equal on the recorded fixtures).  For tests: it exists
 * so that what the fast body of ebm_hmc_chain_f32 trades away (merged kicks, fused multiply-adds, the hoisted eps / m) is a measured
 * quantity; it costs 2 L + 2 evaluations per transition where the fast body spends L.  EBM_EKIND for other energies.
 */
EBM_API int ebm_hmc_chain_audit_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                                    int32_t n_mh, int32_t n_leapfrog, float eps, const float* eps_table,
                                    int32_t mass_kind, double mass_scalar, const float* mass_diag, int32_t thin,
                                    float* traj, uint8_t* accept_mask, uint32_t* accept_count,
                                    const float* p_noise, const float* u, uint64_t seed, uint64_t offset, void* stream);

/* The accept step with the RNG coordinates in DEVICE memory (rng_state = {seed, step}; the uniforms
 * are drawn at step rng_state[1] + step_delta): the graph-capturable form, see
 * ebm_langevin_step_dev_f32.  No injected-uniform form. */
EBM_API int ebm_hmc_accept_dev_f32(float* x, const float* x_prop, const float* h0, const float* h1,
                           uint8_t* accept_mask, uint32_t* accept_count, int64_t n_chains,
                           int32_t dim, const uint64_t* rng_state, uint64_t step_delta, void* stream);

/*
 * Noise-free descent samplers (reference: torchebm/samplers/gradient_descent.py).
 * torch.sub / torch.add with `alpha` are single-rounding FMAs on the CPU reference, and so are these:
 *   gradient descent (:121-123):   x' = fma(-eta, g(x), x)
 *   Nesterov (:262-266):           la = fma(mu, v, x);  v' = fma(-eta, g(la), fl(v*mu));  x' = x + v'
 * ebm_descent_chain_f32 runs k fused steps for an analytic energy (v starts at zero and is not
 * returned, as in the reference); eta_table = NULL or device float[k]; traj as in the Langevin chain.
 * ebm_descent_step_f32 is one update with an external gradient (v == NULL: plain descent; else v is
 * updated in place); ebm_lookahead_f32 forms the Nesterov look-ahead point.  Outputs may alias x.
 */
EBM_API int ebm_descent_chain_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                                  int32_t k_steps, float eta, const float* eta_table, int32_t nesterov,
                                  float momentum, int32_t thin, float* traj, void* stream);
EBM_API int ebm_descent_step_f32(const float* x, const float* grad, float* v, float* out, int64_t n_elem,
                                 float eta, float momentum, void* stream);
EBM_API int ebm_lookahead_f32(const float* x, const float* v, float* out, int64_t n_elem, float momentum,
                              void* stream);

/*
 * Persistent-CD replay-buffer traffic in one launch each (reference:
 * torchebm/core/base_loss.py:296-315 stratified read, :390-426 FIFO write).
 *   gather:  row_i = (i*stride + r_i) % buffer_size,  r_i uniform in {0..stride-1};  out[i,:] = buffer[row_i,:]
 *            r_i = offsets[i] when offsets != NULL (injected), else floor(o_i * stride / 2^32) with o_i the
 *            raw 32-bit output of the native RNG field at step `offset`, element i.  rows_out (optional)
 *            receives row_i.  stride = buffer_size / batch (>= 1).
 *   scatter: buffer[(write_pos + i) % buffer_size, :] = samples[i, :]   for i < batch <= buffer_size
 */
EBM_API int ebm_pcd_gather_f32(const float* buffer, int64_t buffer_size, int32_t dim, float* out, int64_t batch,
                               int64_t stride, const int64_t* offsets, int64_t* rows_out,
                               uint64_t seed, uint64_t offset, void* stream);
EBM_API int ebm_pcd_scatter_f32(float* buffer, int64_t buffer_size, int32_t dim, const float* samples,
                                int64_t batch, int64_t write_pos, void* stream);

/*
 * ABI 6 -- the three launches of a persistent-CD TRAINING STEP with every per-step coordinate in DEVICE memory, so that the
 * whole step (start points, the k-fused chain, the FIFO write, the caller's loss / backward / optimiser) can be captured in
 * ONE HIP graph and replayed with fresh draws and an advancing write position: what the host would otherwise bake into the
 * kernel arguments at capture time lives in two small device buffers that launches inside the graph read and that a
 * one-element add inside the same graph advances (torchebm_amd/utils/graphed_step.py; reference call sites:
 * torchebm/losses/contrastive_divergence.py:82-155, core/base_loss.py:266-337,390-426).  Same convention as
 * ebm_langevin_step_dev_f32 / ebm_noise_fill_dev_f32: rng_state = {seed, step}; the launch draws at step rng_state[1] + step_delta.
 *   ebm_langevin_chain_dev_f32: ebm_langevin_chain_f32 with native draws (no injected noise, no diagnostics records) at steps
 *       rng_state[1] + step_delta .. + k_steps - 1.  Energies: EBM_ENERGY_MLP (every shape the chain kernel takes); EBM_EKIND
 *       for the others (their kernels take the coordinates by value).
 *   ebm_pcd_gather_dev_f32: ebm_pcd_gather_f32 with native offsets drawn at step rng_state[1] + step_delta.
 *   ebm_pcd_scatter_dev_f32: ebm_pcd_scatter_f32 with the write position read from *write_pos (device int64, 0 <= *write_pos
 *       < buffer_size; advancing it is the caller's: (pos + batch) % buffer_size).
 */
EBM_API int ebm_langevin_chain_dev_f32(const ebm_energy_t* energy, float* x, int64_t n_chains, int32_t dim,
                                       int32_t k_steps, float eta, float sqrt_eta, float noise_coef,
                                       const float* coef_table, int32_t clamp_on, float cmin, float cmax,
                                       int32_t thin, float* traj, const uint64_t* rng_state, uint64_t step_delta,
                                       void* stream);
EBM_API int ebm_pcd_gather_dev_f32(const float* buffer, int64_t buffer_size, int32_t dim, float* out, int64_t batch,
                                   int64_t stride, int64_t* rows_out, const uint64_t* rng_state, uint64_t step_delta,
                                   void* stream);
EBM_API int ebm_pcd_scatter_dev_f32(float* buffer, int64_t buffer_size, int32_t dim, const float* samples,
                                    int64_t batch, const int64_t* write_pos, void* stream);

/*
 * ABI 7 -- the chain starts of a persistent-CD step in ONE launch: the stratified gather of ebm_pcd_gather_f32 (row i of out = buffer
 * row i stride + U_i, U_i uniform in [0, stride)) AND the reference's exploration noise on a random subset of exactly n_noise rows
 * (core/base_loss.py:316-332: start_points[randperm(batch)[:n_new]] += 0.01 randn; n_noise = max(1, int(batch new_sample_ratio)),
 * noise_scale = 0.01 there).  The subset is { i : pi(i) < n_noise } for a keyed pseudo-random bijection pi of [0, batch) (six-round
 * Feistel network on ceil(log2 batch) bits with cycle walking; round keys from the Philox field): every subset of that size
 * (pseudo-)equally likely, as randperm's prefix is -- no sort, no index list, no torch-side draw, so the step replays from a HIP
 * graph with the same numbers as the eager loop.  Consumes THREE steps of the field: step (offsets, as the gather), step + 1 (round
 * keys: groups 0, 1), step + 2 (the normals, element e of out).  rng_state (optional): {seed, step0} in device memory, `step` is then
 * an offset from step0 and `seed` is ignored (the _dev form of the other entries).  batch < 2^31.  n_noise = 0: the plain gather.
 */
EBM_API int ebm_pcd_start_points_f32(const float* buffer, int64_t buffer_size, int32_t dim, float* out, int64_t batch,
                                     int64_t stride, int64_t n_noise, float noise_scale, uint64_t seed, uint64_t step,
                                     const uint64_t* rng_state, void* stream);

/*
 * ABI 7 -- the contrastive-divergence loss of ONE model call on [data | negatives] (losses/contrastive_divergence.py:128-155):
 *   L = mean(E+) - mean(E-) + reg (mean(E+^2) + mean(E-^2)),   e_both = float[2 n] = E+ | E-;
 * a non-finite L becomes the constant 0.1 and sends no gradient (:150-155).  ebm_cd_loss_f32: one launch -- block partials of the
 * four sums in fp64, added in block order by the last block to finish -- writes loss_out[1] and finite_out[1] (1.0 / 0.0).
 * work: device memory of ebm_cd_loss_work_bytes() bytes, 8-byte aligned, ZEROED once by the caller (the ticket counter in it is left
 * at zero again by every launch, so consecutive calls on one stream share it).  ebm_cd_loss_backward_f32: the per-row seed
 * seed_out[2 n] = dL/dE_i = (+-1 + 2 reg E_i) / n * upstream[0] * finite[0] (upstream: dL_total/dL, a device scalar) -- what
 * ebm_mlp_param_grads_f32 takes as `seed`.  torch's graph of the same arithmetic is some twenty launches of 4 - 5 us.
 */
EBM_API int64_t ebm_cd_loss_work_bytes(void);
EBM_API int ebm_cd_loss_f32(const float* e_both, int64_t n, float reg, void* work, float* loss_out, float* finite_out, void* stream);
EBM_API int ebm_cd_loss_backward_f32(const float* e_both, int64_t n, float reg, const float* upstream, const float* finite,
                                     float* seed_out, void* stream);

/* Energy E(x)[n_chains] and gradient dE/dx[n_chains, dim] of a fused analytic energy
 * (either output may be NULL).  core/base_model.py:143-148,181-210,224-229. */
EBM_API int ebm_energy_grad_f32(const ebm_energy_t* energy, const float* x, int64_t n_chains,
                        int32_t dim, float* energy_out, float* grad_out, void* stream);

/*
 * ABI 6 / 7 -- the forward + backward of a training step through an EBM_ENERGY_MLP network, for the parameter gradients (what autograd
 * does for loss.backward() through the energies of torchebm/losses/contrastive_divergence.py:128-155; the network of
 * examples/20-training/01-mcmc-losses/02-persistent-cd/main.py:21-31).  One launch evaluates the network on x[n, dim], runs the
 * backward through it for a UNIT seed on the matrix cores, everything on-chip, and stores the three activation planes the parameter
 * gradients are made of, in TILES of 32 rows, hidden-major within a tile:
 *   acts = float[n_pad / 32][3][H][32],  n_pad = n rounded up to a multiple of 128 (tile t = rows 32 t .. 32 t + 31, one contiguous
 *   block of 12 H floats -- what one wavefront writes and what one step of ebm_mlp_param_grads_f32 reads; rows n .. n_pad - 1 are
 *   written too, with the values of an all-zero input row: the gradient pass gives them seed 0):
 *   [.][0] h1 = silu(W1 x + b1)   [.][1] a2 = W2 h1 + b2   [.][2] d1 = (W2^T (w3 silu'(a2))) silu'(a1)
 * (ABI 6 stored h1 | seed h2 | seed d2 | seed d1 as [4][H][n_pad]; h2 = silu(a2) and d2 = w3 silu'(a2) are functions of a2 and w3 that the
 * gradient pass recomputes, a quarter of the bytes less).  From these   dW2 = d2 h1^T,  db2 = d2 1,  dW1 = d1 x,  db1 = d1 1,
 * dw3 = h2 1,  db3 = 1   -- each summand weighted with its row's seed -- are small-output products over K = n
 * (ebm_mlp_param_grads_f32 below makes them in one pass).  energy_out (optional): E(x)[n];  grad_out (optional): seed dE/dx [n, dim]
 * (seed[n], NULL = 1, scales grad_out ONLY: the planes are seed-free).  The autograd graph of the same step materialises a1, h1, a2,
 * h2 and their gradients -- some forty passes over [n, H] arrays; this is one.  Hidden width 64 or 128, dim <= 64 (EBM_EDIM
 * otherwise).  With energy_out set it IS the training forward; without a gradient to prepare the forward is ebm_energy_grad_f32 with
 * grad_out = NULL, which runs the forward pass only.
 */
EBM_API int ebm_mlp_backward_acts_f32(const ebm_energy_t* energy, const float* x, int64_t n_chains, int32_t dim,
                                      const float* seed, float* energy_out, float* grad_out, float* acts, void* stream);

/*
 * ABI 7 -- the parameter gradients themselves, from the planes ebm_mlp_backward_acts_f32 stored (acts = float[n_pad / 32][3][H][32]:
 * h1 | a2 | d1) and the rows x[n_rows, dim] they were made from, in ONE pass over the planes:
 *   grads_out = float[H dim + H + H H + H + H + 1], the packed parameter order of EBM_ENERGY_MLP:  dW1 | db1 | dW2 | db2 | dw3 | db3.
 * seed (optional, [n_rows]; NULL = 1): the per-row dL/dE, applied HERE, on load (d1, d2 and h2 are linear in it), so that the ONE
 * ebm_mlp_backward_acts_f32 launch of the forward pass serves the backward pass too.  w3: the last layer's weights [H] (d2 = w3
 * silu'(a2); the packed parameter block + H dim + H + H H + H).  h2 and d2 are recomputed from a2 with the forward kernel's own
 * arithmetic (hardware exp2 / rcp).  The products run on v_mfma_f32_32x32x2_f32 (exact fp32 products, fp32 accumulation); every
 * workgroup writes one partial record into `work` and a second kernel adds the records in a fixed order: same inputs, same bits,
 * whatever the scheduling.  work: device float[work_floats], work_floats >= ebm_mlp_param_grads_work_f32(hidden, dim, n_rows) (a
 * per-device figure: two records per CU; 0 for an unsupported shape).  Hidden width 64 or 128, dim <= 64 (EBM_EDIM otherwise).
 * Reference: what autograd does for loss.backward() through the nn.Linear weights (torchebm/losses/contrastive_divergence.py:128-155).
 */
EBM_API int64_t ebm_mlp_param_grads_work_f32(int32_t hidden, int32_t dim, int64_t n_rows);
EBM_API int ebm_mlp_param_grads_f32(const float* acts, int64_t n_rows, int32_t hidden, const float* x, int32_t dim, const float* seed,
                                    const float* w3, float* work, int64_t work_floats, float* grads_out, void* stream);

/* Column statistics for the sampler diagnostics (samplers/langevin_dynamics.py:173-185):
 * mean[dim], biased var[dim] clamped to [1e-10, 1e10].  `work` = device double[2*dim + 1], zeroed once by
 * the caller: the kernel's last block finishes the statistics and leaves it zeroed again, so consecutive
 * calls on one stream share it without a memset. */
EBM_API int ebm_chain_stats_f32(const float* x, int64_t n_chains, int32_t dim, float* mean_out,
                        float* var_out, double* work, void* stream);

/* Fill `out[n_elem]` with the native RNG field at step `offset` (tests / verification /
 * BaseSampler._init_state-style draws): normals, uniforms in [0,1), or raw u32 bits. */
EBM_API int ebm_noise_fill_f32(float* out, int64_t n_elem, int32_t kind, uint64_t seed,
                       uint64_t offset, void* stream);

/* The same field with {seed, step} read from DEVICE memory, at step rng_state[1] + step_delta (the
 * momentum draw of a HMC transition captured in a HIP graph, samplers/hmc.py:92-134). */
EBM_API int ebm_noise_fill_dev_f32(float* out, int64_t n_elem, int32_t kind, const uint64_t* rng_state,
                           uint64_t step_delta, void* stream);

/*
 * In-kernel sampler diagnostics (SURVEY.md section 8b `diag_partials`, section 5 "kernels emit per-block partial sums").
 * At every kept step each workgroup of a chain launch stores ONE record of its chains' per-column partials
 * (sum, M2 about the block mean), its energy sum and its accept count; ebm_diag_finish_f32 merges the records of
 * all kept steps into the reference's diagnostics tensors.  return_diagnostics=True is therefore one chain launch
 * plus one small merge launch, with no extra pass over the state.  (The matrix-layout MLP kernels store one record per
 * WAVEFRONT of 32 chains -- the state lives in the MFMA accumulator layout, a column statistic is a sum over 32 lanes, no
 * LDS tile -- and write a Langevin record's energy share one evaluation late, when the energy of the kept state exists;
 * a kept last step costs one forward pass more.  Since ABI version 3.)
 *
 * ebm_diag_layout: the record geometry the chain entry WOULD use for this energy / shape --
 *   sampler: EBM_DIAG_LANGEVIN / EBM_DIAG_LANGEVIN_HEUN / EBM_DIAG_HMC;  injected_noise / with_traj: whether the
 *   chain call will pass a noise / trajectory pointer (they select the kernel family);
 *   outputs: n_blocks (records per kept step), slots S and block_elems E.  The caller allocates
 *   diag_partials = float[n_kept][n_blocks][2*S + 8] and work = double[n_kept][3*dim + 3] (zeroed once; the merge
 *   leaves it zeroed; 3*max(S, dim) + 3 doubles per kept step when S > dim, below).  (Dense Gaussians above 128 dims: records from the
 *   streamed-Ps kernels, csrc/gauss_big.hip, for multiples of 4 up to 512.)  S > dim means PACKED rows: a dense
 *   Gaussian whose width the matrix-layout kernel does not take as is (below 20, or not a multiple of 4) runs S / dim
 *   consecutive chains as one row of width S (block-diagonal precision; same element order, same random field), and
 *   the records are those of n_chains * dim / S rows: call ebm_diag_finish_f32 with (n_chains * dim / S, S) and fold
 *   the S columns onto the dim coordinates -- mean = average of the S / dim column means of a coordinate, var = average
 *   of their variances + the (biased) variance of those column means, energy divided by S / dim
 *   (torchebm_amd/samplers/langevin.py, _fused_with_records).
 *   block_elems < 0 (since ABI version 5) means records of INTERLEAVED ALIGNMENT CLASSES: the shifted-row kernels (dense
 *   Gaussians and mixtures at widths that are not a multiple of 4, 17 .. 254) give a workgroup the chains of ONE alignment
 *   class -- K = 4 / gcd(dim, 4) classes (4 for an odd width, 2 for width = 2 mod 4) -- so record b = (group b / K, class b % K)
 *   holds the 32 chains 32 K g + K m + s, m = 0 .. 31, of group g = b / K and class s = b % K; block_elems = -32 * dim,
 *   slots = dim, and n_blocks = ceil(n_chains / (32 K)) * K.  Treat the value as opaque: allocate diag_partials with the
 *   n_blocks and slots returned and hand block_elems to ebm_diag_finish_f32 unchanged (it accepts the negative form); a caller
 *   that computes with block_elems itself must take |block_elems| / dim = 32 chains per record and the interleaving above.
 *   Returns EBM_EDIM / EBM_EKIND when the configuration has no in-kernel form (then take the
 *   statistics from the state with ebm_chain_stats_f32 / ebm_energy_grad_f32 between launches).
 * ebm_diag_finish_f32: mean_out / var_out = float[n_kept][dim] (biased variance clamped to [1e-10, 1e10], zero for a
 *   single chain), energy_out = float[n_kept] (mean per-chain energy), accept_out = NULL or float[n_kept]
 *   (accepted fraction of the kept transition).  fp64 merge by the pairwise-variance identity.
 */
enum { EBM_DIAG_LANGEVIN = 0, EBM_DIAG_LANGEVIN_HEUN = 1, EBM_DIAG_HMC = 2 };
EBM_API int ebm_diag_layout(const ebm_energy_t* energy, int32_t sampler, int64_t n_chains, int32_t dim,
                            int32_t injected_noise, int32_t with_traj, int64_t* n_blocks, int32_t* slots,
                            int32_t* block_elems);
EBM_API int ebm_diag_finish_f32(const float* diag_partials, int32_t n_kept, int64_t n_blocks, int32_t slots,
                                int32_t block_elems, int64_t n_chains, int32_t dim, float* mean_out, float* var_out,
                                float* energy_out, float* accept_out, double* work, void* stream);

/* Measurement aid (bench.py, no counterpart in the reference): `blocks` x 256 lanes each issue
 * 8 * iters independent v_fma_f32.  blocks * 4 * 8 * iters wave-instructions / elapsed time = the plain-VALU
 * issue rate of this chip at its current clock, against which the VALU-bound chain kernels are priced.
 * `out` = float[blocks * 256] (keeps the arithmetic alive). */
EBM_API int ebm_probe_valu_f32(float* out, int32_t blocks, int32_t iters, void* stream);

/* The same stream shape with the instruction classes the in-kernel RNG is made of, so that bench.py can price the
 * Langevin loop from costs measured in the run instead of constants: per lane and iteration,
 *   kind 0: 8 v_fma_f32 (= ebm_probe_valu_f32)          kind 1: 8 x (v_mad_u64_u32 + v_xor_b32)
 *   kind 2: 8 x (v_log_f32 + v_add_f32)                 kind 3: 8 v_pk_fma_f32 (16 fused multiply-adds; SGPR-pair multiplicand)
 *   kind 4: 8 v_bitop3_b32                              kind 5: 8 v_pk_fma_f32, every operand a VGPR pair
 *   kind 6: 8 v_pk_mul_f32
 *   kind 7: the lean Langevin loop's static mix per float4 group and step -- 18 v_mad_u64_u32, 20 v_bitop3_b32, 8 transcendentals,
 *           20 packed-f32 (14 v_pk_mul_f32 + 6 v_pk_add_f32), 10 plain (76 instructions per iteration; `iters` even), dependency-free: the ceiling of a kernel made of that mix
 * independent across the eight slots (`iters` a multiple of 4 for kinds 3 / 5 / 6).  Since ABI version 3. */
EBM_API int ebm_probe_issue_f32(float* out, int32_t blocks, int32_t iters, int32_t kind, void* stream);

/* The pre-split first-layer image of an EBM_ENERGY_MLP network (see the enum: ebm_energy_t.aux).  Both weight matrices as
 * bf16 triples are 192 KB at H = 128 and dim > 64 -- more than a CU's LDS -- so the W1 triple is laid out once in global
 * memory (the caller's buffer, 16-byte aligned, ebm_mlp_w1_image_bytes() long; 0 = this shape has no image) in the order the
 * chain kernel streams it through LDS.  Rebuild it whenever the parameters change (one small kernel; stream-ordered like every
 * entry).  No counterpart in the reference: there the network is nn.Linear modules evaluated by autograd every step
 * (samplers/langevin_dynamics.py:168-172).  Since ABI version 4. */
/* The same for an EBM_ENERGY_GAUSSIAN precision matrix at dims 132 .. 512 (multiples of 4): the three bf16 pieces of P
 * (hi + mid + lo = the fp32 value) in the order the stages of the tiled Langevin kernel consume them, 1.5 x the size of P --
 * and, since ABI version 5, at the widths 158 .. 254 that are not a multiple of 4: one image per alignment class of the
 * shifted rows (K = 4 / gcd(dim, 4) of them, each laid out for the row shifted by that class's offset), K x the size.
 * `prec` is the SYMMETRIC matrix handed over as dev1.  0 bytes = this width has no image.  Since ABI version 4. */
EBM_API size_t ebm_gauss_prec_image_bytes(int32_t dim);
EBM_API int ebm_gauss_prec_image_f32(const float* prec, int32_t dim, void* image, void* stream);

/* The active-column hint of EBM_ENERGY_GMM (its `aux`), computed on the device in ONE small launch (since ABI version 5):
 * out[0] = sum_v 2^v * [ the component means differ somewhere in columns 4v..4v+3 ],  means [n_comp, dim], dim % 4 == 0,
 * dim <= 32.  Replaces the seven tensor ops the Python model spent on it at every sample() call (45 us of launches in
 * front of a 0.1 - 1 ms kernel).  `!=` as IEEE: a NaN mean differs from everything. */
EBM_API int ebm_gmm_active_columns_i32(const float* means, int32_t n_comp, int32_t dim, int32_t* out, void* stream);

EBM_API size_t ebm_mlp_w1_image_bytes(int32_t hidden, int32_t dim);
EBM_API int ebm_mlp_w1_image_f32(const float* params, int32_t hidden, int32_t dim, void* image, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* EBM_HIP_H */
