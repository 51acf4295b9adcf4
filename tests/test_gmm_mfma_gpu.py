"""Gaussian-mixture HMC on the matrix-layout kernel (csrc/gauss_hmc_mfma.hip: GmmE): mixtures of up to 32 components
at dims 20 .. 128 (96 with a diagonal mass) run the two K x dim passes of the gradient on the bf16 matrix pipe with
three-way split operands.  Its own Philox draws against the oracle fed with the same field, over component counts on
both sides of the 8 / 16 / 32 register classes, every tile count and every mass form; the energy it reports in the
diagnostics-free path is checked through the accept decisions (bit-identical to the oracle's wherever the oracle's own
margin is not within round-off)."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib, _rng

pytestmark = pytest.mark.gpu


def _field(shape, seed, steps, device, kind=None):
    kind = _lib.NOISE_NORMAL if kind is None else kind
    rows = []
    for st in steps:
        buf = torch.empty(shape, device=device)
        _lib.call("ebm_noise_fill_f32", buf.data_ptr(), buf.numel(), kind, seed, st, _lib.stream_handle(device))
        rows.append(buf)
    return torch.stack(rows)


@pytest.mark.parametrize("K", [1, 3, 8, 9, 16, 17, 32])
@pytest.mark.parametrize("dim,mass", [(20, None), (32, None), (32, 1.7), (36, "diag"), (64, None), (64, "diag"), (96, None), (96, 0.6),
                                      (96, "diag"), (100, None), (128, 1.3)])
def test_mixture_hmc_matrix_kernel_matches_oracle(cuda_device, K, dim, mass):
    g = torch.Generator().manual_seed(100 * K + dim)
    means = torch.randn(K, dim, generator=g) * 1.2
    weights = torch.rand(K, generator=g) + 0.2
    model = ta.GaussianMixtureModel(means, sigma=1.1, weights=weights, device=cuda_device)
    en = oracle.GaussianMixture(means, 1.1, log_weights=model.log_weights.detach().cpu())
    if mass == "diag":
        mass = torch.rand(dim, generator=g) + 0.5
    n, T, L, thin, eps = 97, 4, 6, 2, 0.12
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L,
                                 mass=mass.to(cuda_device) if torch.is_tensor(mass) else mass, device=cuda_device)
    x0 = torch.randn(n, dim, generator=g).clamp_(-2.0, 2.0)
    seed = 7000 + K + dim
    traj = s.sample(x=x0.to(cuda_device), n_steps=T, thin=thin, return_trajectory=True,
                    generator=torch.Generator(device=cuda_device).manual_seed(seed))
    p = _field((n, dim), _rng.kernel_seed(seed), range(0, 2 * T, 2), cuda_device).cpu()
    u = _field((n,), _rng.kernel_seed(seed), range(1, 2 * T, 2), cuda_device, kind=_lib.NOISE_UNIFORM).cpu()
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, mass=mass, thin=thin, want_traj=True)
    assert torch.isfinite(traj).all() and traj.shape == want["trajectory"].shape
    err = ((traj.cpu() - want["trajectory"]).abs() / want["trajectory"].abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
    # per chain: a chain that sits near a tie between components turns fp32 round-off of the logits into a visibly
    # different responsibility (reference and kernel are equally far from exact arithmetic there) -- such a chain may be
    # off by more, never by much; typical chains agree to ~1e-6
    if want["margin"] > 1e-4:
        assert (err <= 5e-4).float().mean().item() >= 0.97 and (err <= 5e-3).all(), err.max().item()
        assert err.median().item() <= 2e-5
    else:
        assert (err <= 5e-4).float().mean().item() >= 0.9


def test_dim_32_small_mixtures_stay_on_the_lane_group_kernel(cuda_device):
    """dim 32 with K <= 8 keeps one lane per chain (means as scalar operands, the active-column body for the ring): the ring
    and a dense mixture of that shape against the oracle, and a K = 9 mixture of the same width on the matrix kernel."""
    n, dim, T, L, eps = 512, 32, 5, 20, 0.1
    for name in ("ring", "dense8", "dense9"):
        g = torch.Generator().manual_seed(3)
        if name == "ring":
            model = ta.core.ring_mixture(8, dim, device=cuda_device)
            en = oracle.GaussianMixture(model.means.detach().cpu().float(), 1.0)
        else:
            means = torch.randn(int(name[5:]), dim, generator=g) * 1.5
            model, en = ta.GaussianMixtureModel(means, sigma=1.0, device=cuda_device), oracle.GaussianMixture(means, 1.0)
        s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, device=cuda_device)
        x0 = torch.randn(n, dim, generator=g)
        c0 = hip_calls("ebm_hmc_chain_f32")
        out = s.sample(x=x0.to(cuda_device), n_steps=T, generator=torch.Generator(device=cuda_device).manual_seed(11))
        assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
        p = _field((n, dim), _rng.kernel_seed(11), range(0, 2 * T, 2), cuda_device).cpu()
        u = _field((n,), _rng.kernel_seed(11), range(1, 2 * T, 2), cuda_device, kind=_lib.NOISE_UNIFORM).cpu()
        want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, want_traj=False)
        err = ((out.cpu() - want["x"]).abs() / want["x"].abs().clamp(min=1.0)).amax(dim=1)
        assert (err <= 5e-4).float().mean().item() >= 0.97, (name, err.max().item())


@pytest.mark.parametrize("K", [1, 5, 9, 16, 25, 32])
@pytest.mark.parametrize("dim", [20, 32, 64, 96, 128])
def test_mixture_langevin_matrix_kernel(cuda_device, K, dim):
    """Langevin on mixtures in the matrix layout (csrc/gauss_mfma.hip with gmm3::Mixture): native draws == the same field
    injected (bit for bit, clamp + schedule + thinned trajectory included), and the injected run against the oracle."""
    from torchebm_amd.samplers.langevin import em_coefficients

    g = torch.Generator().manual_seed(10 * K + dim)
    means = torch.randn(K, dim, generator=g) * 1.3
    model = ta.GaussianMixtureModel(means, sigma=0.9, device=cuda_device)
    en = oracle.GaussianMixture(means, 0.9)
    spec = model.fused_spec()
    n, k, thin = 130, 9, 3
    etas = [0.03 * 0.93 ** i for i in range(k)]
    sigs = [1.0 - 0.02 * i for i in range(k)]
    rows = [em_coefficients(e, s) for e, s in zip(etas, sigs)]
    table = torch.tensor([(r[0], r[1], r[2], 0.0) for r in rows], dtype=torch.float32, device=cuda_device)
    x0 = torch.randn(n, dim, generator=g).clamp_(-2, 2)
    seed = _rng.kernel_seed(900 + K + dim)

    def run(noise, sd):
        x = x0.to(cuda_device).clone()
        traj = torch.full((n, k // thin, dim), float("nan"), device=cuda_device)
        _lib.call("ebm_langevin_chain_f32", spec.to_c(), x.data_ptr(), n, dim, k, rows[0][0], rows[0][1], rows[0][2], table.data_ptr(),
                  1, -2.2, 2.4, thin, traj.data_ptr(), None, _lib.ptr(noise), sd, 0, _lib.stream_handle(cuda_device))
        return x, traj

    xa, ta_ = run(None, seed)
    noise = _field((n, dim), seed, range(k), cuda_device)
    xb, tb = run(noise, 0)
    assert torch.equal(xa, xb) and torch.equal(ta_, tb)
    want_x, want_t, _ = oracle.langevin_chain(en, x0, noise.cpu(), etas, sigs, clamp=(-2.2, 2.4), thin=thin, want_traj=True)
    err = ((tb.cpu() - want_t).abs() / want_t.abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
    assert (err <= 5e-5).float().mean().item() >= 0.95 and (err <= 5e-3).all(), err.max().item()
