"""the negatives of Energy Matching -- two Langevin calls per training step, one
under a TemperatureScheduler sweep (sigma_i = 0 below tau_star, then sqrt(eps(t_i))), one at a
constant sqrt(eps_max) -- and the contrastive term built on them.

Fixtures ``tests/golden/em_*.pt`` were written by the reference's ``EnergyMatchingLoss._sample_negatives``
(tests/golden/make_golden.py) together with every draw it consumed.  CPU tests: the package class
reproduces them bit for bit from the seed alone (same draw order), and the oracle reproduces them from
the stored draws.  GPU tests: the fused kernel, fed the stored draws and the sweep as its per-step
coefficient table, reproduces the reference bit for bit (element-wise energies); through the public
class each call is one launch."""

import math
import os

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import golden_names, hip_calls, load_golden, oracle_energy, package_model
from torchebm_amd.core.schedules import ConstantScheduler, TemperatureScheduler
from torchebm_amd.losses import EnergyMatchingContrastive, trimmed_mean

CASES = golden_names("em_")


def _loss(fx, model, device=None, **kw):
    return EnergyMatchingContrastive(
        model, lambda_cd=2.0, epsilon_max=fx["epsilon_max"], tau_star=fx["tau_star"], n_langevin_steps=fx["k"],
        langevin_dt=fx["dt"], noise_fraction=fx["noise_fraction"], device=device, **kw)


def test_fixtures_present():
    assert len(CASES) == 3


@pytest.mark.parametrize("name", CASES)
def test_class_reproduces_reference_negatives_from_the_seed(name):
    fx = load_golden(name)
    model = package_model(fx["energy"])
    loss = _loss(fx, model)
    out = loss(fx["x1"], generator=torch.Generator().manual_seed(fx["seed"]))
    assert torch.equal(out["negatives"], fx["ref"]["negatives"])
    assert torch.equal(out["cd_value"], fx["ref"]["cd_value"])
    assert torch.equal(out["cd_loss"], fx["ref"]["cd_loss"])
    assert not out["negatives"].requires_grad
    # the loss leaves the constant (or sweep) scheduler registered on the sampler it owns, like the reference
    sched = loss.sampler.schedulers["noise_scale"]
    assert isinstance(sched, ConstantScheduler if fx["n"] > fx["n_noise"] else TemperatureScheduler)


@pytest.mark.parametrize("name", CASES)
def test_oracle_reproduces_reference_negatives_from_the_stored_draws(name):
    fx = load_golden(name)
    energy = oracle_energy(fx["energy"])
    k, parts = fx["k"], []
    if fx["n_noise"] > 0:
        sig = fx["sigma_sweep"]
        assert sig[0] == 0.0 and all(b >= a for a, b in zip(sig, sig[1:])) and sig[-1] < math.sqrt(fx["epsilon_max"])
        parts.append(oracle.langevin_chain(energy, fx["init"], fx["noise_sweep"], [fx["dt"]] * k, sig)[0])
    if fx["n"] > fx["n_noise"]:
        const = [math.sqrt(fx["epsilon_max"])] * k
        parts.append(oracle.langevin_chain(energy, fx["x1"][fx["pick"]], fx["noise_const"], [fx["dt"]] * k, const)[0])
    assert torch.equal(torch.cat(parts), fx["ref"]["negatives"])


def test_sweep_schedule_values():
    s = TemperatureScheduler(epsilon_max=0.15, tau_star=0.6, n_steps=24)
    assert s.preview(24) == load_golden("em_dw_61x4")["sigma_sweep"]


def test_validation_warmup_phase_and_trimmed_mean():
    model = ta.DoubleWellModel()
    for bad in ({"noise_fraction": 1.5}, {"cd_trim_fraction": 1.0}, {"cd_clamp": -1.0}, {"langevin_dt": 0.0}):
        with pytest.raises(ValueError):
            EnergyMatchingContrastive(model, **bad)
    with pytest.raises(ValueError):
        trimmed_mean(torch.arange(4.0), 1.0)
    v = torch.tensor([5.0, 1.0, 3.0, 100.0, 2.0])
    assert trimmed_mean(v, 0.0).item() == v.mean().item()
    assert trimmed_mean(v, 0.2).item() == 2.75 and trimmed_mean(v, 0.41).item() == 2.0
    loss = EnergyMatchingContrastive(model, lambda_cd=0.0, n_langevin_steps=4)
    out = loss(torch.randn(8, 2))
    assert set(out) == {"cd_loss"} and out["cd_loss"].item() == 0.0          # no chains in the warm-up phase
    loss.lambda_cd = 2.0
    out = loss(torch.randn(8, 2))
    assert out["negatives"].shape == (8, 2) and out["cd_loss"].item() >= -0.02
    # gradient reaches the potential's parameters through both energies, not through the chains
    net = ta.MLPEnergy(2)
    out = EnergyMatchingContrastive(net, n_langevin_steps=3, cd_clamp=None)(torch.randn(16, 2))
    out["cd_loss"].backward()
    assert all(p.grad is not None for p in net.parameters())


def test_conditioning_rows_follow_their_chains():
    """Batch-aligned conditioning is sliced per part and re-assembled in negative order."""

    class Shifted(ta.BaseModel):
        def forward(self, x, shift=None):
            return 0.5 * ((x - shift) ** 2).sum(-1)

    model = Shifted()
    x1 = torch.randn(10, 2)
    shift = torch.arange(10.0).unsqueeze(1).expand(10, 2).contiguous()
    loss = EnergyMatchingContrastive(model, n_langevin_steps=2, noise_fraction=0.3)
    neg = loss.sample_negatives(x1, model_kwargs={"shift": shift}, generator=torch.Generator().manual_seed(0))
    aligned = loss._neg_model_kwargs["shift"]
    assert neg.shape == (10, 2) and aligned.shape == (10, 2)
    assert torch.equal(aligned[:3], shift[:3])
    g = torch.Generator().manual_seed(0)
    torch.randn(3, 2, generator=g)
    for _ in range(2):
        torch.randn(3, 2, generator=g)
    pick = torch.randperm(10, generator=g)[:7]
    assert torch.equal(aligned[3:], shift[pick])


@pytest.mark.skipif(not os.path.isdir("/root/reference/torchebm"), reason="reference checkout not present")
def test_reference_loss_runs_unmodified_on_this_sampler():
    """The reference's EnergyMatchingLoss, imported as is, drives this package's LangevinDynamics
    (register_scheduler + sample) and gets the negatives it gets from its own sampler."""
    import sys
    import types

    sys.dont_write_bytecode = True
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    if "torchebm._version" not in sys.modules:
        v = types.ModuleType("torchebm._version")
        v.__version__ = "0.0.0+reference"
        sys.modules["torchebm._version"] = v
    from torchebm.core import DoubleWellModel as RefDoubleWell
    from torchebm.losses import EnergyMatchingLoss

    fx = load_golden("em_dw_61x4")
    ref_model = RefDoubleWell(barrier_height=fx["energy"]["h"], b=fx["energy"]["b"])
    mine = ta.LangevinDynamics(ref_model, step_size=fx["dt"], noise_scale=1.0)
    loss = EnergyMatchingLoss(model=ref_model, sampler=mine, epsilon_max=fx["epsilon_max"], tau_star=fx["tau_star"],
                              n_langevin_steps=fx["k"], noise_fraction=fx["noise_fraction"])
    neg = loss._sample_negatives(fx["x1"], generator=torch.Generator().manual_seed(fx["seed"]))
    assert torch.equal(neg, fx["ref"]["negatives"])


# ------------------------------------------------------------------------------------------
# GPU
# ------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
def test_fused_kernel_with_sweep_table_is_bit_exact(cuda_device, name):
    from test_langevin_gpu import _chain_call
    from torchebm_amd.samplers.langevin import em_coefficients

    fx = load_golden(name)
    spec = package_model(fx["energy"], device=cuda_device).fused_spec()
    k, parts = fx["k"], []
    if fx["n_noise"] > 0:
        x = fx["init"].to(cuda_device).clone()
        rows = [em_coefficients(fx["dt"], s) for s in fx["sigma_sweep"]]
        _chain_call(spec, x, k, rows, None, 1, None, fx["noise_sweep"].to(cuda_device).contiguous())
        parts.append(x.cpu())
    if fx["n"] > fx["n_noise"]:
        x = fx["x1"][fx["pick"]].to(cuda_device).clone()
        rows = [em_coefficients(fx["dt"], math.sqrt(fx["epsilon_max"]))]
        _chain_call(spec, x, k, rows, None, 1, None, fx["noise_const"].to(cuda_device).contiguous())
        parts.append(x.cpu())
    assert torch.equal(torch.cat(parts), fx["ref"]["negatives"])


@pytest.mark.gpu
def test_class_on_gpu_two_fused_launches_and_statistics(cuda_device):
    """Through the public class on an analytic energy each Langevin call is ONE launch, the sweep's
    sigma = 0 prefix is deterministic gradient flow, and the constant-temperature chains sit at the
    Boltzmann density exp(-V / eps_max) of the harmonic potential: var = eps_max / k."""
    eps_max, kspring, n = 0.2, 1.5, 1 << 16
    model = ta.HarmonicModel(k=kspring, device=cuda_device)
    loss = EnergyMatchingContrastive(model, epsilon_max=eps_max, tau_star=0.5, n_langevin_steps=600, langevin_dt=0.01,
                                     noise_fraction=0.5, device=cuda_device)
    x1 = torch.randn(n, 4, device=cuda_device) * math.sqrt(eps_max / kspring)
    c0 = hip_calls("ebm_langevin_chain_f32")
    out = loss(x1, generator=torch.Generator(device=cuda_device).manual_seed(3))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 2
    neg = out["negatives"]
    assert neg.shape == x1.shape and torch.isfinite(neg).all() and torch.isfinite(out["cd_loss"])
    const_part = neg[n // 2:]
    assert abs(const_part.var().item() - eps_max / kspring) < 0.03 * eps_max / kspring
    assert abs(const_part.mean().item()) < 5e-3
    # the same seed gives the same negatives; a sweep that never leaves sigma = 0 is noise-free
    again = loss(x1, generator=torch.Generator(device=cuda_device).manual_seed(3))["negatives"]
    assert torch.equal(again, neg)
    flow = EnergyMatchingContrastive(model, epsilon_max=0.0, n_langevin_steps=50, noise_fraction=1.0, device=cuda_device)
    start = torch.randn(256, 4, device=cuda_device)
    a = flow.sample_negatives(start, x0=start, generator=torch.Generator(device=cuda_device).manual_seed(1))
    want = start * (1.0 - 0.01 * kspring) ** 50
    assert torch.allclose(a.sort(dim=0).values, want.sort(dim=0).values, rtol=1e-4, atol=1e-6)
