"""SURVEY.md §8(f) n3, Heun-SDE Langevin: the fused ``ebm_langevin_heun_chain_f32`` kernel (two gradient
evaluations per step, in-kernel noise, clamp, thinning) against the reference's recorded
``LangevinDynamics(integrator="heun")`` runs with the noise it consumed, and through the public class.

Bars: element-wise energies bit-exact (the halves of the Heun average are exact, the sum rounds once,
every other op is the Euler-Maruyama update already proven bit-exact); Gaussian / mixture |dx| <= 2e-5."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import golden_names, hip_calls, load_golden, oracle_energy, package_model
from torchebm_amd import _lib, _rng
from torchebm_amd.samplers.langevin import em_coefficients

pytestmark = pytest.mark.gpu


def _heun_call(spec, x, k, rows, clamp, thin, traj, noise, seed=0, step=0):
    n, dim = x.shape
    a, sq, coef = rows[0]
    table = None
    if len(rows) > 1:
        table = torch.tensor([(r[0], r[1], r[2], 0.0) for r in rows], dtype=torch.float32, device=x.device)
    clamp_on, cmin, cmax = (0, 0.0, 0.0) if clamp is None else (1, clamp[0], clamp[1])
    _lib.call(
        "ebm_langevin_heun_chain_f32", spec.to_c(), x.data_ptr(), n, dim, k, a, sq, coef, _lib.ptr(table),
        clamp_on, cmin, cmax, thin, _lib.ptr(traj), None, _lib.ptr(noise), seed, step, _lib.stream_handle(x.device),
    )


@pytest.mark.parametrize("name", golden_names("heun_"))
def test_heun_kernel_injected_noise_matches_reference(cuda_device, name):
    fx = load_golden(name)
    spec = package_model(fx["energy"], device=cuda_device).fused_spec()
    x = fx["x0"].to(cuda_device).clone()
    k, thin = fx["k"], fx["thin"]
    rows = [em_coefficients(e, s) for e, s in zip(fx["etas"], fx["sigmas"])]
    if len(set(rows)) == 1:
        rows = rows[:1]
    traj = torch.full((fx["n"], k // thin, fx["dim"]), float("nan"), device=cuda_device)
    _heun_call(spec, x, k, rows, fx["clamp"], thin, traj, fx["noise"].to(cuda_device).contiguous())
    if fx["energy"]["kind"] in ("double_well", "harmonic"):
        assert torch.equal(x.cpu(), fx["ref"]["x"])
        assert torch.equal(traj.cpu(), fx["ref"]["trajectory"])
    else:
        assert (x.cpu() - fx["ref"]["x"]).abs().max().item() <= 2e-5
        assert (traj.cpu() - fx["ref"]["trajectory"]).abs().max().item() <= 2e-5


@pytest.mark.parametrize("kind", ["dw", "gmm", "gauss"])
def test_sampler_heun_fused_route_native_rng(cuda_device, kind):
    """integrator="heun" on an analytic energy is ONE launch; its Philox field is the Euler-Maruyama
    chain's (step s, element e), so the oracle fed with ebm_noise_fill_f32's output reproduces it."""
    n, dim, k, eta = 300, 6, 9, 0.02
    x0 = torch.randn(n, dim).clamp_(-2.0, 2.0)
    if kind == "dw":
        model, en = ta.DoubleWellModel(device=cuda_device), oracle.DoubleWell()
    elif kind == "gmm":
        means = torch.randn(4, dim, generator=torch.Generator().manual_seed(3)) * 1.5
        model, en = ta.GaussianMixtureModel(means, sigma=0.9, device=cuda_device), oracle.GaussianMixture(means, 0.9)
    else:
        a = torch.randn(dim, dim, generator=torch.Generator().manual_seed(4))
        mean, cov = torch.randn(dim, generator=torch.Generator().manual_seed(5)), a @ a.t() / dim + 0.5 * torch.eye(dim)
        model, en = ta.GaussianModel(mean, cov, device=cuda_device), oracle.Gaussian(mean, cov)
    s = ta.LangevinDynamics(model, step_size=eta, noise_scale=0.8, clamp=(-3.0, 3.0), integrator="heun", device=cuda_device)
    h0, c0 = hip_calls("ebm_langevin_heun_chain_f32"), hip_calls("ebm_langevin_chain_f32")
    gen = torch.Generator(device=cuda_device).manual_seed(21)
    got = s.sample(x=x0.to(cuda_device), n_steps=k, generator=gen)
    assert hip_calls("ebm_langevin_heun_chain_f32") == h0 + 1 and hip_calls("ebm_langevin_chain_f32") == c0
    assert gen.get_offset() == 4 * k
    noise = []
    for i in range(k):
        buf = torch.empty(n, dim, device=cuda_device)
        _lib.call("ebm_noise_fill_f32", buf.data_ptr(), n * dim, _lib.NOISE_NORMAL, _rng.kernel_seed(21), i, _lib.stream_handle(cuda_device))
        noise.append(buf)
    want, _, _ = oracle.langevin_chain(en, x0, torch.stack(noise).cpu(), [eta] * k, [0.8] * k, clamp=(-3.0, 3.0), integrator="heun")
    if kind == "dw":
        assert torch.equal(got.cpu(), want)
    else:
        torch.testing.assert_close(got.cpu(), want, rtol=3e-5, atol=3e-5)
    # trajectory + diagnostics path (one launch per kept step)
    traj, diag = s.sample(x=x0.to(cuda_device), n_steps=6, thin=2, return_trajectory=True, return_diagnostics=True)
    assert traj.shape == (n, 3, dim) and torch.isfinite(diag["energy"]).all() and traj.abs().max().item() <= 3.0


def test_heun_is_second_order_in_the_drift(cuda_device):
    """Noise-free check of what Heun buys: for dx/dt = -k x the one-step error against exp(-k h) is
    O(h^3) for Heun and O(h^2) for Euler; sigma -> tiny isolates the drift update."""
    kspring, h = 1.5, 0.1
    model = ta.HarmonicModel(k=kspring, device=cuda_device)
    x0 = torch.linspace(-2, 2, 4096, device=cuda_device).reshape(-1, 4).contiguous()
    em = ta.LangevinDynamics(model, step_size=h, noise_scale=1e-12, device=cuda_device)
    heun = ta.LangevinDynamics(model, step_size=h, noise_scale=1e-12, integrator="heun", device=cuda_device)
    exact = x0 * torch.exp(torch.tensor(-kspring * h * 10))
    err_em = (em.sample(x=x0, n_steps=10) - exact).abs().max().item()
    err_heun = (heun.sample(x=x0, n_steps=10) - exact).abs().max().item()
    assert err_heun < err_em / 10


def test_heun_with_mlp_energy_keeps_the_eager_loop(cuda_device):
    """No fused Heun kernel for the MLP energy: the C ABI refuses it, the sampler runs eager torch ops."""
    model = ta.MLPEnergy(2, device=cuda_device)
    spec = model.fused_spec()
    x = torch.randn(64, 2, device=cuda_device)
    with pytest.raises(RuntimeError):
        _heun_call(spec, x, 2, [em_coefficients(0.01, 1.0)], None, 1, None, None)
    s = ta.LangevinDynamics(model, step_size=0.01, integrator="heun", device=cuda_device)
    h0 = hip_calls("ebm_langevin_heun_chain_f32")
    out = s.sample(x=x, n_steps=3)  # (warns once per process that this runs eager torch ops)
    assert hip_calls("ebm_langevin_heun_chain_f32") == h0 and torch.isfinite(out).all()
