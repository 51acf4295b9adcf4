"""Host-side logic on CPU: API contract, CPU route against the golden fixtures, schedulers,
integrator plug-in rules, generator contract, contrastive divergence.  No GPU needed.
Modelled on the reference's tests/samplers/test_api_contract.py, test_langevin_dynamics.py,
test_hmc.py, tests/test_generator.py and tests/losses/test_contrastive_divergence.py."""

import inspect
import math
import os

import pytest
import torch

import torchebm_amd as ta
from helpers import golden_names, load_golden, package_model
from torchebm_amd import _lib
from torchebm_amd.core import (
    BaseSDERungeKuttaIntegrator,
    BaseSymplecticIntegrator,
    ConstantScheduler,
    CosineScheduler,
    ExponentialDecayScheduler,
    LinearScheduler,
    MultiStepScheduler,
    TemperatureScheduler,
    WarmupScheduler,
)
from torchebm_amd.integrators import get_integrator, resolve_integrator

torch.set_num_threads(1)
EXACT = ("double_well", "harmonic")


def _check(got, want, kind):
    if kind in EXACT:
        assert torch.equal(got, want)
    else:
        torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)


# ---------------------------------------------------------------------------------------
# sample() contract
# ---------------------------------------------------------------------------------------
EXPECTED_PARAMS = [
    ("x", None), ("dim", None), ("n_steps", 100), ("n_samples", 1), ("thin", 1),
    ("return_trajectory", False), ("return_diagnostics", False), ("reset_schedulers", True),
]


@pytest.mark.parametrize("cls", [ta.LangevinDynamics, ta.HamiltonianMonteCarlo])
def test_sample_signature(cls):
    sig = inspect.signature(cls.sample)
    params = list(sig.parameters.values())[1:]
    for p, (name, default) in zip(params, EXPECTED_PARAMS):
        assert p.name == name and p.default == default and p.kind == p.POSITIONAL_OR_KEYWORD
    tail = {p.name: p for p in params[len(EXPECTED_PARAMS):]}
    assert set(tail) == {"model_kwargs", "generator"}
    assert all(p.kind == p.KEYWORD_ONLY and p.default is None for p in tail.values())


def test_constructor_signatures_and_validation():
    # the reference's parameters, in the reference's order, `integrator` last (the reference's own test_api_contract pins that); this
    # package's switches (round 6: fused_arithmetic, exact; donate_input) are attributes, default off
    names = list(inspect.signature(ta.LangevinDynamics.__init__).parameters)[1:]
    assert names == ["model", "step_size", "noise_scale", "decay", "clamp", "dtype", "device", "integrator"]
    names = list(inspect.signature(ta.HamiltonianMonteCarlo.__init__).parameters)[1:]
    assert names == ["model", "step_size", "n_leapfrog_steps", "mass", "dtype", "device", "integrator"]
    assert ta.LangevinDynamics.fused_arithmetic is False and ta.HamiltonianMonteCarlo.exact is False
    m = ta.DoubleWellModel()
    with pytest.raises(ValueError, match="step_size must be positive"):
        ta.LangevinDynamics(m, step_size=0.0)
    with pytest.raises(ValueError, match="noise_scale must be positive"):
        ta.LangevinDynamics(m, noise_scale=-1.0)
    with pytest.raises(ValueError, match="clamp min must be < max"):
        ta.LangevinDynamics(m, clamp=(1.0, 0.0))
    with pytest.raises(ValueError, match="n_leapfrog_steps must be positive"):
        ta.HamiltonianMonteCarlo(m, n_leapfrog_steps=0)
    s = ta.LangevinDynamics(m)
    with pytest.raises(ValueError, match="thin must be >= 1"):
        s.sample(dim=2, thin=0)
    with pytest.raises(ValueError, match="dim must be provided"):
        s.sample()
    with pytest.raises(ValueError, match="dim must be provided"):
        ta.HamiltonianMonteCarlo(m).sample(n_samples=3, n_steps=1)


def test_return_types_thin_and_tuple_dim():
    s = ta.LangevinDynamics(ta.DoubleWellModel(), step_size=0.01)
    out = s.sample(dim=3, n_samples=5, n_steps=7)
    assert isinstance(out, torch.Tensor) and out.shape == (5, 3)
    traj = s.sample(dim=3, n_samples=5, n_steps=7, thin=2, return_trajectory=True)
    assert traj.shape == (5, 3, 3)
    out, diag = s.sample(dim=3, n_samples=5, n_steps=7, thin=3, return_diagnostics=True)
    assert out.shape == (5, 3) and diag["mean"].shape == (2, 3) and diag["var"].shape == (2, 3) and diag["energy"].shape == (2,)
    assert s.sample(dim=(3,), n_samples=4, n_steps=2).shape == (4, 3)

    class Flat(ta.core.BaseModel):  # a model that accepts [B, 2, 3] states
        def forward(self, x):
            return 0.5 * x.flatten(1).pow(2).sum(-1)

    out = ta.LangevinDynamics(Flat(), step_size=0.01).sample(dim=(2, 3), n_samples=4, n_steps=2)
    assert out.shape == (4, 2, 3)
    one, diag = s.sample(dim=3, n_samples=1, n_steps=2, return_diagnostics=True)
    assert torch.equal(diag["var"], torch.zeros(2, 3))
    h = ta.HamiltonianMonteCarlo(ta.GaussianModel(torch.zeros(2), torch.eye(2)), step_size=0.1)
    out, diag = h.sample(n_samples=6, n_steps=4, thin=2, return_diagnostics=True)
    assert out.shape == (6, 2) and set(diag) == {"mean", "var", "energy", "acceptance_rate"}
    assert diag["acceptance_rate"].shape == (2,)


# ---------------------------------------------------------------------------------------
# CPU route (BASELINE config 1 plumbing) against the reference's recorded outputs
# ---------------------------------------------------------------------------------------
def _sched(values):
    """Rebuild a scheduler object producing the recorded per-step values (constant or table)."""
    if len(set(values)) == 1:
        return values[0]

    class Table(ta.core.BaseScheduler):
        def __init__(self, vals):
            super().__init__(vals[0])
            self.vals = list(vals)

        def _compute_value(self):
            return self.vals[min(self.step_count, len(self.vals) - 1)]

    return Table(values)


@pytest.mark.parametrize("name", golden_names("ld_"))
def test_langevin_cpu_route_reproduces_reference(name):
    fx = load_golden(name)
    kind = fx["energy"]["kind"]
    s = ta.LangevinDynamics(package_model(fx["energy"]), step_size=_sched(fx["etas"]), noise_scale=_sched(fx["sigmas"]),
                            clamp=fx["clamp"])
    gen = torch.Generator().manual_seed(fx["run_seed"])
    x0 = fx["x0"].clone()
    out = s.sample(x=x0, n_steps=fx["k"], generator=gen)
    _check(out, fx["ref"]["x"], kind)
    assert torch.equal(x0, fx["x0"])
    traj, diag = s.sample(x=x0, n_steps=fx["k"], thin=fx["thin"], return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator().manual_seed(fx["run_seed"]))
    _check(traj, fx["ref"]["trajectory"], kind)
    for key in ("mean", "var", "energy"):
        torch.testing.assert_close(diag[key], fx["ref"]["diagnostics"][key], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("name", golden_names("hmc_"))
def test_hmc_cpu_route_reproduces_reference(name):
    fx = load_golden(name)
    kind = fx["energy"]["kind"]
    s = ta.HamiltonianMonteCarlo(package_model(fx["energy"]), step_size=_sched(fx["eps"]), n_leapfrog_steps=fx["L"],
                                 mass=fx["mass"])
    out, diag = s.sample(x=fx["x0"].clone(), n_steps=fx["T"], thin=1, return_diagnostics=True,
                         generator=torch.Generator().manual_seed(fx["run_seed"]))
    _check(out, fx["ref"]["x"], kind)
    assert torch.equal(diag["acceptance_rate"], fx["ref"]["acceptance_rate_all"])


def test_config1_checksum():
    """BASELINE config 1: LangevinDynamics on the 2-D Gaussian, n=1024, k=100, CPU."""
    fx = load_golden("survey_ld_gauss2d")
    s = ta.LangevinDynamics(ta.GaussianModel(torch.zeros(2), torch.eye(2)), step_size=0.01, noise_scale=1.0)
    x = s.sample(dim=2, n_samples=1024, n_steps=100, generator=torch.Generator().manual_seed(0))
    torch.testing.assert_close(x, fx["ref"]["x"], rtol=2e-5, atol=2e-5)
    fx = load_golden("survey_ld_dw")
    s = ta.LangevinDynamics(ta.DoubleWellModel(), step_size=0.01, noise_scale=1.0)
    x = s.sample(dim=64, n_samples=512, n_steps=50, generator=torch.Generator().manual_seed(123))
    assert torch.equal(x, fx["ref"]["x"])


def test_integrators_cpu_known_answers():
    fx = load_golden("integrators")
    drift = lambda x_, t_: -(x_**3)  # noqa: E731
    em, lf = ta.EulerMaruyamaIntegrator(), ta.LeapfrogIntegrator()
    x, p = fx["x"], fx["p"]
    assert torch.equal(em.step({"x": x}, 0.01, drift=drift, noise=fx["noise"], noise_scale=0.7)["x"], fx["em_sde"])
    assert torch.equal(em.step({"x": x}, 0.01, drift=drift)["x"], fx["em_ode"])
    out = lf.step({"x": x, "p": p}, 0.05, 2.5, drift=drift)
    assert torch.equal(out["x"], fx["lf_step_mass"]["x"]) and torch.equal(out["p"], fx["lf_step_mass"]["p"])
    out = lf.integrate({"x": x, "p": p}, 0.05, 7, fx["mass_t"], drift=drift, safe=True)
    assert torch.equal(out["x"], fx["lf_int_mass_t_safe"]["x"]) and torch.equal(out["p"], fx["lf_int_mass_t_safe"]["p"])
    out = lf.step({"x": fx["bad_x"], "p": p}, 0.05, drift=drift, safe=True)
    assert torch.equal(out["x"], fx["lf_step_safe_bad"]["x"])
    assert torch.equal(x, fx["x"]) and torch.equal(p, fx["p"])
    with pytest.raises(ValueError, match="drift must be provided"):
        lf.step({"x": x, "p": p}, 0.05)
    with pytest.raises(ValueError, match="n_steps must be positive"):
        lf.integrate({"x": x, "p": p}, 0.05, -1, drift=drift)
    with torch.inference_mode():
        a = lf.integrate({"x": x, "p": p}, 0.05, 3, drift=drift)
    b = lf.integrate({"x": x, "p": p}, 0.05, 3, drift=drift, inference_mode=True)
    assert torch.equal(a["x"], b["x"])


# ---------------------------------------------------------------------------------------
# integrator plug-in rules (reference tests/samplers/test_langevin_dynamics.py:252-299)
# ---------------------------------------------------------------------------------------
def test_integrator_resolution_rules():
    m = ta.DoubleWellModel()
    assert isinstance(ta.LangevinDynamics(m).integrator, ta.EulerMaruyamaIntegrator)
    assert isinstance(ta.LangevinDynamics(m, integrator="euler_maruyama").integrator, ta.EulerMaruyamaIntegrator)
    inst = ta.EulerMaruyamaIntegrator(device=torch.device("cpu"), dtype=torch.float32)
    assert ta.LangevinDynamics(m, integrator=inst).integrator is inst
    with pytest.raises(TypeError, match="requires a BaseSDERungeKuttaIntegrator"):
        ta.LangevinDynamics(m, integrator=ta.LeapfrogIntegrator(device=torch.device("cpu"), dtype=torch.float32))
    with pytest.raises(TypeError, match="requires a BaseSDERungeKuttaIntegrator"):
        ta.LangevinDynamics(m, integrator="leapfrog")
    with pytest.raises(ValueError, match="does not match"):
        ta.LangevinDynamics(m, integrator=ta.EulerMaruyamaIntegrator(dtype=torch.float64))
    with pytest.raises(ValueError, match="Unknown integrator"):
        ta.LangevinDynamics(m, integrator="nope")
    with pytest.raises(ValueError, match="ODE/flow family"):
        get_integrator("dopri5")
    assert isinstance(ta.HamiltonianMonteCarlo(m).integrator, ta.LeapfrogIntegrator)

    class NonSeparable(BaseSymplecticIntegrator):
        separable = False

        def step(self, state, step_size, *a, **k):
            return state

        def integrate(self, state, step_size, n_steps, *a, **k):
            return state

    with pytest.raises(TypeError, match="separable"):
        ta.HamiltonianMonteCarlo(m, integrator=NonSeparable(device=torch.device("cpu"), dtype=torch.float32))
    r = resolve_integrator(None, default="leapfrog", family=BaseSymplecticIntegrator, owner="X",
                           device=torch.device("cpu"), dtype=torch.float32)
    assert isinstance(r, ta.LeapfrogIntegrator) and not isinstance(r, BaseSDERungeKuttaIntegrator)


def test_custom_integrator_instance_is_used_on_the_eager_route():
    calls = []

    class Counting(ta.EulerMaruyamaIntegrator):
        def step(self, state, step_size, **kw):
            calls.append(step_size)
            return super().step(state, step_size, **kw)

    s = ta.LangevinDynamics(ta.HarmonicModel(), step_size=0.1, integrator=Counting(device=torch.device("cpu"), dtype=torch.float32))
    s.sample(dim=2, n_samples=3, n_steps=4)
    assert calls == [0.1] * 4


# ---------------------------------------------------------------------------------------
# generator contract (reference tests/test_generator.py:74-113)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("make", [
    lambda: ta.LangevinDynamics(ta.DoubleWellModel(), step_size=0.01),
    lambda: ta.HamiltonianMonteCarlo(ta.DoubleWellModel(), step_size=0.05, n_leapfrog_steps=3),
])
def test_generator_contract_cpu(make):
    s = make()
    a = s.sample(dim=4, n_samples=32, n_steps=5, generator=torch.Generator().manual_seed(1))
    b = s.sample(dim=4, n_samples=32, n_steps=5, generator=torch.Generator().manual_seed(1))
    c = s.sample(dim=4, n_samples=32, n_steps=5, generator=torch.Generator().manual_seed(2))
    assert torch.equal(a, b) and not torch.equal(a, c)
    torch.manual_seed(7)
    d1 = s.sample(dim=4, n_samples=32, n_steps=5)
    torch.manual_seed(7)
    d2 = s.sample(dim=4, n_samples=32, n_steps=5)
    assert torch.equal(d1, d2)
    torch.manual_seed(7)
    state = torch.get_rng_state()
    s.sample(dim=4, n_samples=32, n_steps=5, generator=torch.Generator().manual_seed(3))
    assert torch.equal(state, torch.get_rng_state())  # explicit generator leaves the global RNG alone


# ---------------------------------------------------------------------------------------
# schedulers and pre-expansion
# ---------------------------------------------------------------------------------------
def _walk(s, k):
    out = []
    for _ in range(k):
        out.append(s.get_value())
        s.step()
    return out


@pytest.mark.parametrize("make", [
    lambda: ConstantScheduler(0.3),
    lambda: ExponentialDecayScheduler(1.0, 0.9, 0.5),
    lambda: LinearScheduler(1.0, 0.0, 5),
    lambda: CosineScheduler(0.1, 0.001, 7),
    lambda: MultiStepScheduler(0.1, [2, 5], 0.1),
    lambda: WarmupScheduler(CosineScheduler(0.1, 0.01, 6), 3, 0.01),
    lambda: TemperatureScheduler(0.5, 0.4, 8),
])
def test_preview_and_advance_match_stepping(make):
    a, b = make(), make()
    assert a.preview(12) == _walk(b, 12)
    assert a.step_count == 0  # preview did not touch it
    a.advance(12)
    assert a.step_count == b.step_count == 12 and a.get_value() == b.get_value()
    a.reset()
    assert a.step_count == 0 and a.get_value() == make().get_value()
    c = make()
    c.load_state_dict(b.state_dict())
    assert c.get_value() == b.get_value() and c.step_count == 12


def test_scheduler_formulas_and_validation():
    assert _walk(LinearScheduler(1.0, 0.0, 5), 7) == pytest.approx([1.0, 0.8, 0.6, 0.4, 0.2, 0.0, 0.0])
    assert _walk(ExponentialDecayScheduler(1.0, 0.9, 0.8), 4) == pytest.approx([1.0, 0.9, 0.81, 0.8])
    assert _walk(MultiStepScheduler(1.0, [1, 3], 0.5), 5) == pytest.approx([1.0, 0.5, 0.5, 0.25, 0.25])
    c = _walk(CosineScheduler(1.0, 0.0, 4), 6)
    assert c[0] == 1.0 and c[2] == pytest.approx(0.5) and c[4] == 0.0 and c[5] == 0.0
    w = _walk(WarmupScheduler(ConstantScheduler(1.0), 4, 0.0), 7)
    assert w == pytest.approx([0.0, 0.25, 0.5, 0.75, 1.0, 1.0, 1.0])
    t = TemperatureScheduler(0.36, tau_star=0.5, n_steps=4)
    assert _walk(t, 5) == pytest.approx([0.0, 0.0, 0.0, math.sqrt(0.18), 0.6])
    for bad in (lambda: ExponentialDecayScheduler(1.0, 1.5), lambda: LinearScheduler(1.0, 0.0, 0),
                lambda: MultiStepScheduler(1.0, [3, 2]), lambda: TemperatureScheduler(-1.0)):
        with pytest.raises(ValueError):
            bad()
    with pytest.raises(TypeError):
        ConstantScheduler("x")


def test_sampler_steps_and_resets_schedulers():
    s = ta.LangevinDynamics(ta.HarmonicModel(), step_size=LinearScheduler(0.1, 0.01, 10))
    s.sample(dim=2, n_samples=2, n_steps=4)
    assert s.schedulers["step_size"].step_count == 4
    s.sample(dim=2, n_samples=2, n_steps=3, reset_schedulers=False)
    assert s.schedulers["step_size"].step_count == 7
    s.sample(dim=2, n_samples=2, n_steps=2)
    assert s.schedulers["step_size"].step_count == 2
    with pytest.raises(KeyError):
        s.get_scheduled_value("nope")


# ---------------------------------------------------------------------------------------
# energies
# ---------------------------------------------------------------------------------------
def test_energy_models_and_fused_specs():
    x = torch.randn(9, 5)
    dw = ta.DoubleWellModel(1.5, 0.7)
    torch.testing.assert_close(dw.gradient(x), 4 * 1.5 * x * (x**2 - 0.49))
    spec = dw.fused_spec()
    assert spec.kind == _lib.ENERGY_DOUBLE_WELL and spec.elementwise and spec.scalars[0] == 1.5 and spec.scalars[1] == 0.7**2
    hm = ta.HarmonicModel(3.0)
    torch.testing.assert_close(hm.gradient(x), 3.0 * x)
    assert hm.fused_spec().scalars[0] == 1.5
    g = ta.GaussianModel(torch.zeros(2), torch.tensor([[2.0, 0.3], [0.3, 1.0]]))
    sp = g.fused_spec()
    assert sp.kind == _lib.ENERGY_GAUSSIAN and torch.equal(sp.dev1, sp.dev1.t())
    with pytest.raises(ValueError):
        g(torch.randn(4, 3))
    with pytest.raises(ValueError):
        ta.GaussianModel(torch.zeros(2, 1), torch.eye(2))
    with pytest.raises(ValueError):
        ta.GaussianModel(torch.zeros(2), torch.zeros(2, 2))
    gm = ta.core.ring_mixture(8, 32)
    assert gm.fused_spec().n_comp == 8 and gm.mean.shape == (32,)
    xx = torch.randn(6, 32)
    r = torch.softmax(gm.log_weights - ((xx[:, None] - gm.means[None]) ** 2).sum(-1) / 2, dim=1)
    torch.testing.assert_close(gm.gradient(xx), (r[:, :, None] * (xx[:, None] - gm.means[None])).sum(1), rtol=1e-5, atol=1e-6)

    class Sub(ta.DoubleWellModel):
        def forward(self, x):
            return super().forward(x) + 1.0

    assert Sub().fused_spec() is None  # not the same function any more: never fused

    class NoGrad(ta.core.BaseModel):
        def forward(self, x):
            return torch.zeros(x.shape[0])

    with pytest.raises(RuntimeError, match="differentiable"):
        NoGrad().gradient(x)

    class BadShape(ta.core.BaseModel):
        def forward(self, x):
            return x.sum()

    with pytest.raises(ValueError, match="expected shape"):
        BadShape().gradient(x)
    assert dw.gradient(x.double()).dtype == torch.float64


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libebm_hip.so")
    with pytest.raises(RuntimeError, match="no CPU/PyTorch fallback"):
        _lib.call("ebm_noise_fill_f32", 16, 4, 0, 0, 0, None)


# ---------------------------------------------------------------------------------------
# contrastive divergence (caller of the path)
# ---------------------------------------------------------------------------------------
class FakeSampler:
    """Duck-typed sampler (reference tests/losses/test_contrastive_divergence.py:112-152)."""

    def __init__(self, shift=0.5):
        self.shift = shift
        self.calls = []

    def sample(self, x=None, n_steps=1, **kw):
        self.calls.append((tuple(x.shape), n_steps, sorted(kw)))
        return (x + self.shift).detach()


class TinyEnergy(ta.core.BaseModel):
    def __init__(self):
        super().__init__()
        self.w = torch.nn.Parameter(torch.tensor([1.0, -0.5]))

    def forward(self, x):
        return (x * self.w).sum(-1) ** 2


def test_cd_loss_and_gradients():
    model, fake = TinyEnergy(), FakeSampler()
    cd = ta.ContrastiveDivergence(model, fake, k_steps=7, energy_reg_weight=0.0)
    x = torch.randn(16, 2)
    loss, neg = cd(x)
    assert fake.calls == [((16, 2), 7, ["generator", "model_kwargs"])]
    assert torch.equal(neg, x + 0.5)
    torch.testing.assert_close(loss, model(x).mean() - model(neg).mean())
    loss.backward()
    assert model.w.grad is not None and torch.isfinite(model.w.grad).all()
    cd2 = ta.ContrastiveDivergence(model, fake, k_steps=1, energy_reg_weight=0.1)
    l2, _ = cd2(x)
    torch.testing.assert_close(l2, loss.detach() + 0.1 * ((model(x) ** 2).mean() + (model(neg) ** 2).mean()))


def test_cd_nonfinite_loss_falls_back():
    class NanEnergy(ta.core.BaseModel):
        def forward(self, x):
            return x.sum(-1) * float("nan")

    loss, _ = ta.ContrastiveDivergence(NanEnergy(), FakeSampler(), k_steps=1)(torch.randn(4, 2))
    assert loss.item() == pytest.approx(0.1)


def test_pcd_buffer_fifo_and_stratified_reads():
    model, fake = TinyEnergy(), FakeSampler(0.0)
    cd = ta.ContrastiveDivergence(model, fake, k_steps=1, persistent=True, buffer_size=12, init_steps=0, new_sample_ratio=0.0)
    x = torch.randn(4, 2)
    g = torch.Generator().manual_seed(0)
    starts = cd.get_start_points(x, generator=g)
    assert cd.buffer_initialized and cd.replay_buffer.shape == (12, 2) and starts.shape == (4, 2)
    assert cd.replay_buffer.abs().max() < 0.1  # 0.01-scale noise
    cd.replay_buffer.copy_(torch.arange(12.0)[:, None].expand(12, 2))
    rows = cd.get_start_points(x, generator=g)[:, 0]
    assert all(3 * i <= rows[i].item() < 3 * (i + 1) for i in range(4))  # one row per stride
    cd.update_buffer(torch.full((5, 2), 100.0))
    assert cd._write_pos == 5 and cd.buffer_ptr.item() == 5 and (cd.replay_buffer[:5] == 100).all()
    cd.update_buffer(torch.full((9, 2), 200.0))  # wraps: rows 5..11 then 0..1
    assert cd._write_pos == 2 and (cd.replay_buffer[5:] == 200).all() and (cd.replay_buffer[:2] == 200).all()
    assert (cd.replay_buffer[2:5] == 100).all()
    cd.update_buffer(torch.arange(40.0).view(20, 2))  # larger than the buffer: keeps the last 12
    assert cd._write_pos == 0 and torch.equal(cd.replay_buffer, torch.arange(40.0).view(20, 2)[-12:])
    sd = cd.state_dict()
    assert "replay_buffer" in sd and "buffer_ptr" in sd
    with pytest.warns(UserWarning, match="smaller than batch size"):
        cd.get_start_points(torch.randn(20, 2))


def test_pcd_training_step_with_real_sampler_cpu():
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(2, 16), torch.nn.SiLU(), torch.nn.Linear(16, 1))

    class Mlp(ta.core.BaseModel):
        def __init__(self):
            super().__init__()
            self.net = net

        def forward(self, x):
            return self.net(x).squeeze(-1)

    model = Mlp()
    sampler = ta.LangevinDynamics(model, step_size=0.1, noise_scale=1.0)
    cd = ta.ContrastiveDivergence(model, sampler, k_steps=3, persistent=True, buffer_size=64, init_steps=2)
    opt = torch.optim.SGD(model.parameters(), lr=1e-2)
    before = [p.detach().clone() for p in model.parameters()]
    loss, neg = cd(torch.randn(32, 2), generator=torch.Generator().manual_seed(1))
    assert neg.shape == (32, 2) and not neg.requires_grad and torch.isfinite(loss)
    opt.zero_grad()
    loss.backward()
    opt.step()
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))


@pytest.mark.skipif(not os.path.isdir("/root/reference/torchebm"), reason="reference checkout not present")
def test_reference_cd_runs_unmodified_on_this_sampler():
    """Drop-in check (authoring container only): the REFERENCE's ContrastiveDivergence drives this
    package's LangevinDynamics and gets the same negatives as with its own sampler."""
    import subprocess
    import sys

    code = r'''
import sys, types
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(0, %r)
v = types.ModuleType("torchebm._version"); v.__version__ = "0"; sys.modules["torchebm._version"] = v
import torch
from torchebm.core import DoubleWellModel as RefDW
from torchebm.losses import ContrastiveDivergence as RefCD
from torchebm.samplers import LangevinDynamics as RefLD
import torchebm_amd as ta
x = torch.randn(64, 4, generator=torch.Generator().manual_seed(0))
ref_model = RefDW()
mine = ta.LangevinDynamics(ta.DoubleWellModel(), step_size=0.01)
theirs = RefLD(ref_model, step_size=0.01)
la, na = RefCD(ref_model, mine, k_steps=5)(x, generator=torch.Generator().manual_seed(1))
lb, nb = RefCD(ref_model, theirs, k_steps=5)(x, generator=torch.Generator().manual_seed(1))
assert torch.equal(na, nb) and torch.equal(la, lb)
print("dropin-ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert "dropin-ok" in out.stdout, out.stderr[-2000:]


@pytest.mark.parametrize("name", golden_names("heun_"))
def test_heun_langevin_reproduces_reference(name):
    """LangevinDynamics(integrator="heun") (reference tests/samplers/test_langevin_dynamics.py:259-268):
    the generic explicit-tableau step against the reference's recorded run."""
    from torchebm_amd.integrators import HeunIntegrator

    fx = load_golden(name)
    eta, sigma = fx["etas"][0], fx["sigmas"][0]
    if name.endswith("_sched"):  # the schedulers tests/golden/make_golden.py used for this case
        eta, sigma = LinearScheduler(0.05, 0.005, 10), ExponentialDecayScheduler(1.0, 0.9, 0.3)
    s = ta.LangevinDynamics(package_model(fx["energy"]), step_size=eta, noise_scale=sigma, clamp=fx["clamp"], integrator="heun")
    assert type(s.integrator) is HeunIntegrator
    out = s.sample(x=fx["x0"].clone(), n_steps=fx["k"], generator=torch.Generator().manual_seed(fx["run_seed"]))
    _check(out, fx["ref"]["x"], fx["energy"]["kind"])
    traj = s.sample(x=fx["x0"].clone(), n_steps=fx["k"], thin=fx["thin"], return_trajectory=True,
                    generator=torch.Generator().manual_seed(fx["run_seed"]))
    _check(traj, fx["ref"]["trajectory"], fx["energy"]["kind"])


def test_mlp_energy_adopts_an_existing_sequential_without_copying():
    from torch import nn

    net = nn.Sequential(nn.Linear(2, 128), nn.SiLU(), nn.Linear(128, 128), nn.SiLU(), nn.Linear(128, 1))
    energy = ta.MLPEnergy.from_sequential(net)
    assert energy.net is net and energy.in_dim == 2 and energy.hidden == 128
    assert {id(p) for p in energy.parameters()} == {id(p) for p in net.parameters()}
    x = torch.randn(7, 2)
    assert torch.equal(energy(x), net(x).squeeze(-1))
    assert energy.gradient(x).shape == (7, 2)
    with pytest.raises(ValueError):
        ta.MLPEnergy.from_sequential(nn.Sequential(nn.Linear(2, 8), nn.ReLU(), nn.Linear(8, 1)))


def test_sample_and_gather_single_process_contract():
    """with one process the result is indexed [world=1, pieces, n // pieces, ...] exactly like the
    multi-rank result, `n % pieces` is validated, and the chains are those of `pieces` sample() calls."""
    from torchebm_amd.utils import sample_and_gather

    s = ta.LangevinDynamics(ta.DoubleWellModel(), step_size=0.01)
    x = torch.randn(12, 3, generator=torch.Generator().manual_seed(0))
    loc, gat = sample_and_gather(s, x, 4, pieces=3, generator=torch.Generator().manual_seed(1))
    assert gat.shape == (1, 3, 4, 3) and torch.equal(gat.reshape(12, 3), loc)
    gen = torch.Generator().manual_seed(1)
    want = torch.cat([s.sample(x=x[4 * i : 4 * (i + 1)], n_steps=4, generator=gen) for i in range(3)])
    assert torch.equal(loc, want)
    with pytest.raises(ValueError, match="equal blocks"):
        sample_and_gather(s, x, 4, pieces=5)
    # the default adapts to the shard size, and every block runs at the same point of the schedules
    odd = torch.randn(7, 3, generator=torch.Generator().manual_seed(2))
    loc, gat = sample_and_gather(s, odd, 4, generator=torch.Generator().manual_seed(1))
    assert gat.shape == (1, 1, 7, 3) and torch.equal(loc, s.sample(x=odd, n_steps=4, generator=torch.Generator().manual_seed(1)))
    from torchebm_amd.core.schedules import LinearScheduler

    sched = ta.LangevinDynamics(ta.DoubleWellModel(), step_size=LinearScheduler(0.02, 0.002, 8))
    loc, _ = sample_and_gather(sched, x, 4, pieces=3, generator=torch.Generator().manual_seed(1))
    assert sched.schedulers["step_size"].get_value() == pytest.approx(LinearScheduler(0.02, 0.002, 8).preview(5)[4])  # advanced by 4, once
    fresh = ta.LangevinDynamics(ta.DoubleWellModel(), step_size=LinearScheduler(0.02, 0.002, 8))
    gen = torch.Generator().manual_seed(1)
    want = []
    for i in range(3):
        fresh.reset_schedulers()
        want.append(fresh.sample(x=x[4 * i : 4 * (i + 1)], n_steps=4, generator=gen))
    assert torch.equal(loc, torch.cat(want))


def test_width_mismatch_raises_like_the_reference():
    """a state narrower / wider than the model's own dimension must raise the reference's
    ValueError (core/base_model.py:185-188) on every sampler, never index the parameters with x.shape[1]."""
    g = ta.GaussianModel(torch.zeros(8), torch.eye(8))
    for sampler in (ta.LangevinDynamics(g, step_size=0.01), ta.HamiltonianMonteCarlo(g, step_size=0.05, n_leapfrog_steps=2),
                    ta.samplers.GradientDescentSampler(g, step_size=0.05)):
        for width in (4, 12):
            with pytest.raises(ValueError, match="expected"):
                sampler.sample(x=torch.zeros(5, width), n_steps=2)
    from torchebm_amd.core.energies import fused_spec_for

    assert fused_spec_for(g, torch.zeros(5, 8), {}) is not None       # the model's own width: fusable
    assert fused_spec_for(g, torch.zeros(5, 4), {}) is None           # narrower: step route -> forward raises
    assert fused_spec_for(g, torch.zeros(5, 12), {}) is None          # wider: never read past mean / P
    assert fused_spec_for(ta.DoubleWellModel(), torch.zeros(2, 5000), {}) is None                         # lane-group kernels cap rows at 1024
    assert fused_spec_for(ta.DoubleWellModel(), torch.zeros(2, 5000), {}, cap_elementwise=False) is not None  # the flat Langevin kernel does not
    spec = g.fused_spec()
    assert spec.dim == 8
    mix = ta.GaussianMixtureModel(torch.zeros(3, 6))
    assert mix.fused_spec().dim == 6 and ta.DoubleWellModel().fused_spec().dim is None


def test_three_d_state_of_an_analytic_energy_raises_like_the_reference():
    """The reference's BaseModel.gradient rejects an energy of shape (n, a) for a [n, a, b] state
    (core/base_model.py:95-99; verified against /root/reference when the fixtures were made): so do we, on
    every route -- such states are never flattened onto the fused kernels."""
    s = ta.LangevinDynamics(ta.DoubleWellModel(), step_size=0.01)
    with pytest.raises(ValueError, match="expected shape"):
        s.sample(n_samples=3, dim=(2, 3), n_steps=2)
    assert s.sample(n_samples=3, dim=(2,), n_steps=2).shape == (3, 2)


def test_datasets_module():
    """torchebm_amd.datasets: the two generators on the path (two-moons = config 5's data, the ring mixture = config 3's
    centres); equal to the reference's tensors for the same seed where the reference is present."""
    from torchebm_amd.datasets import GaussianMixtureDataset, TwoMoonsDataset

    moons = TwoMoonsDataset(n_samples=301, noise=0.05, seed=3)
    assert len(moons) == 301 and moons.get_data().shape == (301, 2) and moons[5].shape == (2,)
    assert torch.equal(moons.get_data(), TwoMoonsDataset(n_samples=301, noise=0.05, seed=3).get_data())
    from torchebm_amd.utils.synthetic import two_moons

    assert torch.equal(moons.get_data(), two_moons(301, 0.05, seed=3))
    before = torch.get_rng_state()
    TwoMoonsDataset(n_samples=10, seed=1)
    assert torch.equal(torch.get_rng_state(), before)  # a seeded dataset leaves the global generator alone
    moons.regenerate(seed=4)
    assert not torch.equal(moons.get_data(), TwoMoonsDataset(n_samples=301, noise=0.05, seed=3).get_data())
    with pytest.raises(IndexError):
        moons[301]
    with pytest.raises(ValueError):
        TwoMoonsDataset(n_samples=0)
    ring = GaussianMixtureDataset(n_samples=2001, n_components=8, std=0.05, radius=4.0, seed=0)
    assert ring.get_data().shape == (2001, 2)
    means = ta.core.ring_mixture(8, 2, radius=4.0).means
    assert torch.allclose(ring.centers(), means.cpu().to(torch.float32), atol=1e-6)
    nearest = torch.cdist(ring.get_data(), ring.centers()).argmin(1)
    assert torch.bincount(nearest, minlength=8).tolist() == [251] + [250] * 7
    with pytest.raises(ValueError):
        GaussianMixtureDataset(n_components=0)
    if os.path.isdir("/root/reference/torchebm"):  # authoring container: the reference's own generators, same seeds
        import subprocess
        import sys

        code = r'''
import sys, types
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(0, %r)
v = types.ModuleType("torchebm._version"); v.__version__ = "0"; sys.modules["torchebm._version"] = v
import torch
from torchebm.datasets import TwoMoonsDataset as RefMoons, GaussianMixtureDataset as RefRing
from torchebm_amd.datasets import TwoMoonsDataset, GaussianMixtureDataset
for seed in (0, 4):
    assert torch.equal(TwoMoonsDataset(301, 0.05, seed=seed).get_data(), RefMoons(301, 0.05, seed=seed).get_data())
    assert torch.equal(GaussianMixtureDataset(2001, 8, 0.05, 4.0, seed=seed).get_data(), RefRing(2001, 8, 0.05, 4.0, seed=seed).get_data())
print("datasets-ok")
''' % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert "datasets-ok" in out.stdout, out.stderr[-2000:]


def test_descriptor_keeps_its_tensors_alive():
    """``model.fused_spec().to_c()``: the C descriptor holds raw pointers, the FusedSpec is a temporary -- the descriptor itself
    must own the tensors (a mixture's ``aux`` hint is allocated per call; freed, the caching allocator gives its block to
    the next allocation before the launch reads it)."""
    import gc
    import weakref

    from torchebm_amd.core.energies import FusedSpec

    aux = torch.zeros(1, dtype=torch.int32)
    dev0 = torch.zeros(8, 4)
    refs = [weakref.ref(aux), weakref.ref(dev0)]
    desc = FusedSpec(_lib.ENERGY_GMM, aux=aux, dev0=dev0, n_comp=8, dim=4).to_c()
    del aux, dev0
    gc.collect()
    assert all(r() is not None for r in refs)
    assert desc.aux == refs[0]().data_ptr() and desc.dev0 == refs[1]().data_ptr()
    del desc
    gc.collect()
    assert all(r() is None for r in refs)


def test_thin_mlp_training_backward_equals_autograd_cpu():
    """core.energies._ThinMLPEnergy (the hand-written parameter-gradient backward MLPEnergy.forward takes on CUDA inputs that need
    no gradient) against autograd through the same network, in float64 on the CPU."""
    from torchebm_amd.core.energies import _ThinMLPEnergy, _tall_gram

    torch.manual_seed(0)
    m = ta.MLPEnergy(3, 64).double()
    x = torch.randn(50, 3, dtype=torch.float64)
    n = m.net
    e = _ThinMLPEnergy.apply(x, n[0].weight, n[0].bias, n[2].weight, n[2].bias, n[4].weight, n[4].bias)
    e2 = n(x).squeeze(-1)
    assert torch.equal(e, e2)
    g = torch.autograd.grad((e ** 2).sum() + e.sum(), list(m.parameters()))
    g2 = torch.autograd.grad((e2 ** 2).sum() + e2.sum(), list(m.parameters()))
    for a, b in zip(g, g2):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=1e-12, atol=1e-12)
    a, b = torch.randn(64, 5, dtype=torch.float64), torch.randn(64, 3, dtype=torch.float64)
    torch.testing.assert_close(_tall_gram(a, b), a.t() @ b)
    # the CPU forward of the energy is the plain network (the fast path is for CUDA inputs)
    assert type(m(x).grad_fn).__name__ != "_ThinMLPEnergyBackward"


def test_graphed_training_step_host_side_rules():
    """utils.GraphedTrainingStep on the CPU: refuses to capture (nothing to capture on), `enabled=False` is the plain training loop,
    the device-coordinate bookkeeping (`_rng.DeviceCoords`) hands out offsets the way `_rng.reserve` advances a generator."""
    import pytest as _pytest

    from torchebm_amd import _rng
    from torchebm_amd.utils import GraphedTrainingStep

    torch.manual_seed(0)
    model = ta.MLPEnergy(2)
    sampler = ta.LangevinDynamics(model, step_size=0.1)
    cd = ta.ContrastiveDivergence(model, sampler, k_steps=3, persistent=True, buffer_size=64, init_steps=0, new_sample_ratio=0.0)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    with _pytest.raises(ValueError, match="CUDA"):
        GraphedTrainingStep(cd, opt)
    with _pytest.raises(ValueError, match="contrastive-divergence"):
        GraphedTrainingStep(model, opt)
    step = GraphedTrainingStep(cd, opt, enabled=False, generator=torch.Generator().manual_seed(3))
    x = torch.randn(32, 2)
    before = [p.detach().clone() for p in model.parameters()]
    losses = [step(x)[0] for _ in range(3)]
    assert all(torch.isfinite(l) for l in losses) and step.replays == 0 and step.calls == 3
    assert any(not torch.equal(a, b) for a, b in zip(before, model.parameters()))
    assert cd._write_pos == (3 * 32) % 64
    # the same three steps by hand
    torch.manual_seed(0)
    m2 = ta.MLPEnergy(2)
    s2 = ta.LangevinDynamics(m2, step_size=0.1)
    cd2 = ta.ContrastiveDivergence(m2, s2, k_steps=3, persistent=True, buffer_size=64, init_steps=0, new_sample_ratio=0.0)
    o2 = torch.optim.Adam(m2.parameters(), lr=1e-3)
    g2 = torch.Generator().manual_seed(3)
    for want in losses:
        loss, _ = cd2(x, generator=g2)
        o2.zero_grad()
        loss.backward()
        o2.step()
        assert torch.equal(loss.detach(), want)
    c = _rng.DeviceCoords(torch.device("cpu"))
    assert [c.take(1), c.take(20), c.take(5)] == [0, 1, 21] and c.taken == 26
    assert _rng.DeviceCoords.as_i64((1 << 64) - 1) == -1 and _rng.DeviceCoords.as_i64(5) == 5


def test_paired_cd_loss_equals_the_op_by_op_form():
    """losses.cd._PairedCDLoss (the analytic value + gradient of the CD loss on the energies of one model call) against the
    reference's op-by-op form under autograd (contrastive_divergence.py:141-155), the non-finite guard included."""
    from torchebm_amd.losses.cd import _PairedCDLoss

    torch.manual_seed(1)
    for reg in (0.0, 0.01):
        e = torch.randn(600, dtype=torch.float64, requires_grad=True)
        n = 300
        got = _PairedCDLoss.apply(e, n, reg)
        (g_got,) = torch.autograd.grad(got * 1.7, e)
        e2 = e.view(2, n)
        want = e2[0].mean() - e2[1].mean() + reg * ((e2[0] ** 2).mean() + (e2[1] ** 2).mean())
        (g_want,) = torch.autograd.grad(want * 1.7, e)
        torch.testing.assert_close(got, want, rtol=1e-14, atol=1e-14)
        torch.testing.assert_close(g_got, g_want, rtol=1e-13, atol=1e-16)
    assert torch.autograd.gradcheck(lambda t: _PairedCDLoss.apply(t, 4, 0.1), torch.randn(8, dtype=torch.float64, requires_grad=True))
    bad = torch.randn(10)
    bad[2] = float("nan")
    bad.requires_grad_(True)
    out = _PairedCDLoss.apply(bad, 5, 0.01)
    assert out.item() == pytest.approx(0.1)
    (gb,) = torch.autograd.grad(out, bad)
    assert (gb[torch.arange(10) != 2] == 0).all()  # the constant sends no gradient (the NaN row's own entry is NaN * 0, as in autograd)


def test_cd_forward_restores_the_sampler_and_model_switches_it_borrows():
    """ContrastiveDivergence.forward lends the sampler its fresh start points (donate_input) and lets the energy pack its parameters once
    (_pack_scope) for the duration of ONE step: both are back to their defaults afterwards, also when the sampler raises; a sampler the
    user already configured with donate_input = True keeps it."""
    torch.manual_seed(0)
    model = ta.MLPEnergy(2, 64)
    sampler = ta.LangevinDynamics(model, step_size=0.05)
    cd = ta.ContrastiveDivergence(model, sampler, k_steps=2, persistent=True, buffer_size=64, init_steps=0)
    x = torch.randn(16, 2)
    seen = {}
    plain_sample = sampler.sample

    def spy(*a, **kw):
        seen["donate"], seen["scope"] = sampler.donate_input, model._pack_scope
        return plain_sample(*a, **kw)

    sampler.sample = spy
    loss, neg = cd(x)
    assert seen["donate"] is True and isinstance(seen["scope"], dict)
    assert sampler.donate_input is False and model._pack_scope is None and torch.isfinite(loss)

    def boom(*a, **kw):
        raise RuntimeError("sampler failed")

    sampler.sample = boom
    with pytest.raises(RuntimeError, match="sampler failed"):
        cd(x)
    assert sampler.donate_input is False and model._pack_scope is None
    sampler.sample = plain_sample
    sampler.donate_input = True  # the user's own choice survives
    cd(x)
    assert sampler.donate_input is True
