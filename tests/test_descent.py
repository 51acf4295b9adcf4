"""Noise-free descent samplers: oracle and CPU route against the reference's
recorded outputs (CPU), HIP fused / per-step kernels against the same fixtures (GPU)."""

import inspect

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import golden_names, hip_calls, load_golden, oracle_energy, package_model
from test_host_api import _sched
from torchebm_amd.samplers import GradientDescentSampler, NesterovSampler

torch.set_num_threads(1)
NAMES = golden_names("gd_") + golden_names("nag_")
EXACT = ("double_well", "harmonic")


def _check(got, want, kind, tol=2e-5):
    if kind in EXACT:
        assert torch.equal(got, want)
    else:
        torch.testing.assert_close(got, want, rtol=tol, atol=tol)


def _make(fx, device=None):
    model = package_model(fx["energy"], device=device)
    if fx["momentum"] is None:
        return GradientDescentSampler(model, step_size=_sched(fx["etas"]), device=device)
    return NesterovSampler(model, step_size=_sched(fx["etas"]), momentum=fx["momentum"], device=device)


@pytest.mark.parametrize("name", NAMES)
def test_oracle_reproduces_reference(name):
    fx = load_golden(name)
    x, traj, en = oracle.descent_chain(oracle_energy(fx["energy"]), fx["x0"], fx["etas"], fx["momentum"], fx["thin"],
                                       want_traj=True, want_diag=True)
    kind = fx["energy"]["kind"]
    _check(x, fx["ref"]["x"], kind)
    _check(traj, fx["ref"]["trajectory"], kind)
    torch.testing.assert_close(en, fx["ref"]["diagnostics"]["energy"], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", NAMES)
def test_cpu_route_reproduces_reference(name):
    fx = load_golden(name)
    s = _make(fx)
    kind = fx["energy"]["kind"]
    x0 = fx["x0"].clone()
    _check(s.sample(x=x0, n_steps=fx["k"]), fx["ref"]["x"], kind)
    traj, diag = s.sample(x=x0, n_steps=fx["k"], thin=fx["thin"], return_trajectory=True, return_diagnostics=True)
    _check(traj, fx["ref"]["trajectory"], kind)
    assert set(diag) == {"energy"}
    torch.testing.assert_close(diag["energy"], fx["ref"]["diagnostics"]["energy"], rtol=2e-5, atol=2e-5)
    assert torch.equal(x0, fx["x0"])


def test_contract_and_validation():
    for cls in (GradientDescentSampler, NesterovSampler):
        params = list(inspect.signature(cls.sample).parameters)[1:]
        assert params == ["x", "dim", "n_steps", "n_samples", "thin", "return_trajectory", "return_diagnostics",
                          "reset_schedulers", "model_kwargs", "generator"]
    m = ta.DoubleWellModel()
    with pytest.raises(ValueError, match="momentum must be in"):
        NesterovSampler(m, momentum=1.0)
    with pytest.raises(ValueError, match="step_size must be positive"):
        GradientDescentSampler(m, step_size=-1.0)
    s = GradientDescentSampler(m, step_size=0.01)
    with pytest.raises(ValueError, match="thin must be >= 1"):
        s.sample(dim=2, thin=0)
    with pytest.raises(ValueError, match="dim must be provided"):
        s.sample()
    out = s.sample(dim=3, n_samples=5, n_steps=50)
    assert out.shape == (5, 3)
    # descent goes downhill
    x0 = torch.randn(64, 4)
    assert m(s.sample(x=x0, n_steps=200)).mean() < m(x0).mean()


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_fused_descent_kernel_matches_reference(cuda_device, name):
    fx = load_golden(name)
    s = _make(fx, device=cuda_device)
    kind = fx["energy"]["kind"]
    before = hip_calls("ebm_descent_chain_f32")
    x0 = fx["x0"].to(cuda_device)
    out = s.sample(x=x0, n_steps=fx["k"])
    assert hip_calls("ebm_descent_chain_f32") == before + 1
    _check(out.cpu(), fx["ref"]["x"], kind, tol=5e-5)
    traj, diag = s.sample(x=x0, n_steps=fx["k"], thin=fx["thin"], return_trajectory=True, return_diagnostics=True)
    _check(traj.cpu(), fx["ref"]["trajectory"], kind, tol=5e-5)
    torch.testing.assert_close(diag["energy"].cpu(), fx["ref"]["diagnostics"]["energy"], rtol=5e-5, atol=5e-5)
    assert torch.equal(x0.cpu(), fx["x0"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["gd_dw_64x16", "nag_dw_64x16", "nag_har_20x5_sched"])
def test_per_step_descent_kernels_match_reference(cuda_device, name):
    """A subclassed (non-fusable) model takes the per-step route: autograd gradient + the fused
    update kernel; element-wise energies stay bit-exact."""
    fx = load_golden(name)
    base = type(package_model(fx["energy"]))

    class Sub(base):
        def forward(self, x):
            return super().forward(x)

    spec = fx["energy"]
    model = Sub(barrier_height=spec["h"], b=spec["b"], device=cuda_device) if spec["kind"] == "double_well" else Sub(k=spec["k"], device=cuda_device)
    if fx["momentum"] is None:
        s = GradientDescentSampler(model, step_size=_sched(fx["etas"]), device=cuda_device)
    else:
        s = NesterovSampler(model, step_size=_sched(fx["etas"]), momentum=fx["momentum"], device=cuda_device)
    before = hip_calls("ebm_descent_step_f32")
    out = s.sample(x=fx["x0"].to(cuda_device), n_steps=fx["k"])
    assert hip_calls("ebm_descent_step_f32") == before + fx["k"]
    assert torch.equal(out.cpu(), fx["ref"]["x"])
