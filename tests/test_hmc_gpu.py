"""GPU parity tests for the HMC path: ebm_hmc_chain_f32 (through the C ABI) against the CPU
oracle on the golden fixtures' inputs with injected momentum / uniform draws.

Bar: the accept/reject mask is bit-identical to the oracle's for every
decision whose margin |u - a| exceeds 1e-4 (all of them in the fixtures); states agree to a
float tolerance that reflects L leapfrog steps of fp32 round-off through a different
summation order (|dx| <= 5e-4 * max(1, |x|) for L <= 20)."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import golden_names, hip_calls, load_golden, mass_to, oracle_energy, package_model
from torchebm_amd import _lib
from torchebm_amd.integrators.symplectic import _mass_args

pytestmark = pytest.mark.gpu


def _hmc_call(spec, x, eps_vals, L, mass, thin, traj, mask, counts, p_noise, u, seed=0, step=0, records=None):
    n, dim = x.shape
    T = len(eps_vals)
    table = None
    if len(set(eps_vals)) > 1:
        table = torch.tensor(eps_vals, dtype=torch.float32, device=x.device)
    kind, ms, md = _mass_args(mass, x)
    _lib.call(
        "ebm_hmc_chain_f32", spec.to_c(), x.data_ptr(), n, dim, T, L, eps_vals[0], _lib.ptr(table),
        kind, ms, _lib.ptr(md), thin, _lib.ptr(traj), _lib.ptr(records), _lib.ptr(mask), _lib.ptr(counts),
        _lib.ptr(p_noise), _lib.ptr(u), seed, step, _lib.stream_handle(x.device),
    )


@pytest.mark.parametrize("name", golden_names("hmc_"))
def test_hmc_kernel_injected_noise_matches_oracle(cuda_device, name):
    fx = load_golden(name)
    # expected values: the reference's outputs and the oracle's accept decisions, both stored in
    # the fixture (the oracle was asserted bit-identical to the reference when it was generated)
    want = {"x": fx["ref"]["x"], "trajectory": fx["ref"]["trajectory"], "accepted": fx["accepted"], "margin": fx["margin"]}
    model = package_model(fx["energy"], device=cuda_device)
    spec = model.fused_spec()
    n, dim, T, thin = fx["n"], fx["dim"], fx["T"], fx["thin"]
    x = fx["x0"].to(cuda_device).clone()
    mask = torch.full((T, n), 7, dtype=torch.uint8, device=cuda_device)
    counts = torch.zeros(T, dtype=torch.int32, device=cuda_device)
    traj = torch.full((n, T // thin, dim), float("nan"), device=cuda_device)
    _hmc_call(spec, x, fx["eps"], fx["L"], mass_to(fx["mass"], cuda_device), thin, traj, mask, counts,
              fx["p_noise"].to(cuda_device).contiguous(), fx["u"].to(cuda_device).contiguous())
    got_mask = mask.cpu().bool()
    if name != "hmc_dw_extreme":
        assert want["margin"] > 1e-4, want["margin"]  # fixture decisions are not borderline
    assert torch.equal(got_mask, want["accepted"])
    assert torch.equal(counts.cpu().long(), want["accepted"].sum(dim=1))
    scale = want["x"].abs().clamp(min=1.0)
    assert ((x.cpu() - want["x"]).abs() / scale).max().item() <= 5e-4
    tscale = want["trajectory"].abs().clamp(min=1.0)
    assert ((traj.cpu() - want["trajectory"]).abs() / tscale).max().item() <= 5e-4
    assert torch.isfinite(x).all()


def test_hmc_native_rng_fused_equals_injected(cuda_device):
    """The fused kernel's own Philox draws are the field ebm_noise_fill_f32 materialises:
    momentum at step offset+2t, uniforms at offset+2t+1 -- feeding those back as injected
    noise reproduces the native run bit for bit."""
    n, dim, T, L = 500, 32, 5, 7
    model = ta.core.ring_mixture(8, dim, device=cuda_device)
    spec = model.fused_spec()
    x0 = torch.randn(n, dim, device=cuda_device) * 3
    seed, step0 = 424242, 10
    a = x0.clone()
    mask_a = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
    _hmc_call(spec, a, [0.1] * T, L, None, 1, None, mask_a, None, None, None, seed, step0)
    p = torch.empty(T, n, dim, device=cuda_device)
    u = torch.empty(T, n, device=cuda_device)
    st = _lib.stream_handle(cuda_device)
    for t in range(T):
        _lib.call("ebm_noise_fill_f32", p[t].data_ptr(), n * dim, _lib.NOISE_NORMAL, seed, step0 + 2 * t, st)
        _lib.call("ebm_noise_fill_f32", u[t].data_ptr(), n, _lib.NOISE_UNIFORM, seed, step0 + 2 * t + 1, st)
    b = x0.clone()
    mask_b = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
    _hmc_call(spec, b, [0.1] * T, L, None, 1, None, mask_b, None, p, u)
    assert torch.equal(mask_a, mask_b) and torch.equal(a, b)
    assert 0.5 < mask_a.float().mean().item() <= 1.0


def test_hmc_sampler_api_on_cuda(cuda_device):
    model = ta.DoubleWellModel(device=cuda_device)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=10, device=cuda_device)
    x0 = torch.randn(300, 16, device=cuda_device)
    before = hip_calls("ebm_hmc_chain_f32")
    a = s.sample(x=x0, n_steps=6, generator=torch.Generator(device=cuda_device).manual_seed(1))
    b = s.sample(x=x0, n_steps=6, generator=torch.Generator(device=cuda_device).manual_seed(1))
    c = s.sample(x=x0, n_steps=6, generator=torch.Generator(device=cuda_device).manual_seed(2))
    assert hip_calls("ebm_hmc_chain_f32") == before + 3
    assert torch.equal(a, b) and not torch.equal(a, c)
    traj, diag = s.sample(x=x0, n_steps=9, thin=2, return_trajectory=True, return_diagnostics=True)
    assert traj.shape == (300, 4, 16)
    assert set(diag) == {"mean", "var", "energy", "acceptance_rate"}
    assert diag["acceptance_rate"].shape == (4,) and (diag["acceptance_rate"] > 0.8).all()
    assert torch.isfinite(diag["energy"]).all() and (diag["var"] > 0).all()
    # dim inferred from model.mean
    g = ta.GaussianModel(torch.zeros(3, device=cuda_device), torch.eye(3, device=cuda_device), device=cuda_device)
    out = ta.HamiltonianMonteCarlo(g, step_size=0.2, device=cuda_device).sample(n_samples=50, n_steps=3)
    assert out.shape == (50, 3)
    with pytest.raises(ValueError, match="dim must be provided"):
        s.sample(n_samples=4, n_steps=1)


def test_hmc_step_route_matches_fused_decisions(cuda_device):
    """A non-fusable subclass takes the per-transition route (Philox fill, HIP leapfrog kicks
    around autograd, HIP accept); same generator => same noise; energies differ only by
    summation order, so the samples agree to round-off."""

    class MyWell(ta.DoubleWellModel):
        def forward(self, x):
            return super().forward(x)

    x0 = torch.randn(256, 8, device=cuda_device)
    fused = ta.HamiltonianMonteCarlo(ta.DoubleWellModel(device=cuda_device), step_size=0.05, n_leapfrog_steps=5, mass=1.7,
                                     device=cuda_device)
    stepw = ta.HamiltonianMonteCarlo(MyWell(device=cuda_device), step_size=0.05, n_leapfrog_steps=5, mass=1.7,
                                     device=cuda_device)
    k0 = hip_calls("ebm_leapfrog_kick_f32")
    a = fused.sample(x=x0, n_steps=4, generator=torch.Generator(device=cuda_device).manual_seed(3))
    b = stepw.sample(x=x0, n_steps=4, generator=torch.Generator(device=cuda_device).manual_seed(3))
    # 4 >= GRAPH_MIN_STEPS transitions: the first runs eagerly, ONE is captured behind it (2 x L kick launches pass
    # through the binding) and replayed 3 times
    assert hip_calls("ebm_leapfrog_kick_f32") == k0 + 2 * 5
    stepw.capture_graph = False
    c = stepw.sample(x=x0, n_steps=4, generator=torch.Generator(device=cuda_device).manual_seed(3))
    assert hip_calls("ebm_leapfrog_kick_f32") == k0 + 2 * 5 + 4 * 5 and torch.equal(b, c)  # eager launches: same bits
    rows_equal = ((a - b).abs().max(dim=1).values < 1e-4).float().mean().item()
    assert rows_equal > 0.98  # a borderline accept may flip a chain; everything else matches


def test_leapfrog_integrator_known_answers(cuda_device):
    """LeapfrogIntegrator.step/integrate on CUDA (HIP kick kernels) vs the reference's recorded steps."""
    fx = load_golden("integrators")
    x, p = fx["x"].to(cuda_device), fx["p"].to(cuda_device)
    lf = ta.LeapfrogIntegrator(device=cuda_device)
    drift = lambda x_, t_: -(x_**3)  # noqa: E731
    before = hip_calls("ebm_leapfrog_kick_drift_f32")
    out = lf.step({"x": x, "p": p}, 0.05, drift=drift)
    assert torch.equal(out["x"].cpu(), fx["lf_step"]["x"]) and torch.equal(out["p"].cpu(), fx["lf_step"]["p"])
    out = lf.step({"x": x, "p": p}, 0.05, 2.5, drift=drift)
    assert torch.equal(out["x"].cpu(), fx["lf_step_mass"]["x"]) and torch.equal(out["p"].cpu(), fx["lf_step_mass"]["p"])
    out = lf.integrate({"x": x, "p": p}, 0.05, 7, fx["mass_t"].to(cuda_device), drift=drift, safe=True)
    assert torch.equal(out["x"].cpu(), fx["lf_int_mass_t_safe"]["x"])
    assert torch.equal(out["p"].cpu(), fx["lf_int_mass_t_safe"]["p"])
    out = lf.step({"x": fx["bad_x"].to(cuda_device), "p": p}, 0.05, drift=drift, safe=True)
    assert torch.equal(out["x"].cpu(), fx["lf_step_safe_bad"]["x"]) and torch.equal(out["p"].cpu(), fx["lf_step_safe_bad"]["p"])
    assert hip_calls("ebm_leapfrog_kick_drift_f32") == before + 1 + 1 + 7 + 1
    assert torch.equal(x.cpu(), fx["x"]) and torch.equal(p.cpu(), fx["p"])  # inputs untouched
    with pytest.raises(ValueError, match="n_steps must be positive"):
        lf.integrate({"x": x, "p": p}, 0.05, 0, drift=drift)


def test_hmc_gaussian_statistics_native_rng(cuda_device):
    """reference tests/samplers/test_hmc.py:668-703 (tolerances 0.15/0.25), tightened to 0.05
    with 16k chains."""
    mean = torch.tensor([1.0, -1.0], device=cuda_device)
    cov = torch.tensor([[1.0, 0.5], [0.5, 2.0]], device=cuda_device)
    s = ta.HamiltonianMonteCarlo(ta.GaussianModel(mean, cov, device=cuda_device), step_size=0.15, n_leapfrog_steps=10,
                                 device=cuda_device)
    x, d = s.sample(n_samples=16384, n_steps=60, return_diagnostics=True,
                    generator=torch.Generator(device=cuda_device).manual_seed(11))
    torch.testing.assert_close(x.mean(0), mean, rtol=0.05, atol=0.05)
    torch.testing.assert_close(torch.cov(x.T), cov, rtol=0.08, atol=0.08)
    assert d["acceptance_rate"][-1].item() > 0.9


def test_hmc_full_size_config3(cuda_device):
    """BASELINE config 3 shape (n=2^18, dim=32, L=20, 8-mode mixture): determinism, finiteness,
    acceptance rate near the oracle's (0.99 at eps=0.1), chains stay near a mode ring."""
    n, dim = 1 << 18, 32
    model = ta.core.ring_mixture(8, dim, device=cuda_device)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=20, device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device)
    a, d = s.sample(x=x0, n_steps=4, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(5))
    b = s.sample(x=x0, n_steps=4, generator=torch.Generator(device=cuda_device).manual_seed(5))
    assert torch.equal(a, b) and torch.isfinite(a).all()
    assert (d["acceptance_rate"] > 0.97).all(), d["acceptance_rate"]


def test_step_route_carry_force_halves_the_gradient_calls(cuda_device):
    """``HamiltonianMonteCarlo.carry_force`` (opt-in, step route): the force a leapfrog step ends on starts the next one --
    L + 1 gradient evaluations per transition instead of 2 L, the same chains for a deterministic energy (eager and
    graph-replayed)."""

    class Counting(ta.DoubleWellModel):
        calls = 0

        def forward(self, x):
            return super().forward(x)

        def gradient(self, x, model_kwargs=None):
            type(self).calls += 1
            return super().gradient(x, model_kwargs)

    x0 = torch.randn(3000, 24, device=cuda_device)
    L, T = 6, 5
    outs = {}
    for carry in (False, True):
        for graph in (False, True):
            model = Counting(device=cuda_device)
            assert model.fused_spec() is None
            h = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=L, device=cuda_device)
            h.carry_force, h.capture_graph = carry, graph
            Counting.calls = 0
            outs[carry, graph] = h.sample(x=x0, n_steps=T, generator=torch.Generator(device=cuda_device).manual_seed(3))
            if not graph:
                assert Counting.calls == T * ((L + 1) if carry else 2 * L)
    assert torch.equal(outs[False, False], outs[True, False])
    assert torch.equal(outs[False, False], outs[False, True])
    assert torch.equal(outs[False, False], outs[True, True])


@pytest.mark.parametrize("kind", ["dw", "har"])
@pytest.mark.parametrize("dim,mass", [(9, None), (12, 1.7), (18, None), (24, "diag"), (33, None), (40, 0.6), (48, None), (70, "diag"),
                                      (96, None), (130, 1.3), (192, None), (300, None), (384, "diag"), (600, None), (768, 0.8)])
def test_elementwise_hmc_three_vectors_per_lane(cuda_device, kind, dim, mass):
    """Element-wise energies at row widths in (2^k, 1.5 2^k] float4 vectors run with three vectors per lane (csrc/hmc.hip
    hmc_geometry; csrc/hmc_kernel.h NV == 3): injected draws against the oracle -- accept masks bit for bit, states, kept
    rows -- over the mass forms, widths on and off multiples of 4, with and without records."""
    g = torch.Generator().manual_seed(31 * dim + (kind == "har"))
    n, T, L, thin, eps = 77, 4, 7, 2, 0.07
    if kind == "dw":
        model, en = ta.DoubleWellModel(device=cuda_device), oracle.DoubleWell()
    else:
        model, en = ta.HarmonicModel(device=cuda_device), oracle.Harmonic()
    if mass == "diag":
        mass = torch.rand(dim, generator=g) + 0.5
    x0 = torch.randn(n, dim, generator=g).clamp_(-2.0, 2.0)
    p, u = torch.randn(T, n, dim, generator=g), torch.rand(T, n, generator=g)
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, mass=mass, thin=thin, want_traj=True)
    x = x0.to(cuda_device)
    mask = torch.full((T, n), 7, dtype=torch.uint8, device=cuda_device)
    counts = torch.zeros(T, dtype=torch.int32, device=cuda_device)
    traj = torch.full((n, T // thin, dim), float("nan"), device=cuda_device)
    _hmc_call(model.fused_spec(), x, [eps] * T, L, mass_to(mass, cuda_device), thin, traj, mask, counts,
              p.to(cuda_device), u.to(cuda_device))
    if want["margin"] > 1e-4:
        assert torch.equal(mask.cpu().bool(), want["accepted"])
        assert torch.equal(counts.cpu().long(), want["accepted"].sum(dim=1))
        assert ((x.cpu() - want["x"]).abs() / want["x"].abs().clamp(min=1.0)).max().item() <= 5e-4
        tscale = want["trajectory"].abs().clamp(min=1.0)
        assert ((traj.cpu() - want["trajectory"]).abs() / tscale).max().item() <= 5e-4
    # through the sampler with native draws and records: the diagnostics are those of the trajectory it returns
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, device=cuda_device,
                                 mass=mass.to(cuda_device) if torch.is_tensor(mass) else mass)
    c0 = hip_calls("ebm_hmc_chain_f32")
    tr, diag = s.sample(x=x0.to(cuda_device), n_steps=T, thin=thin, return_trajectory=True, return_diagnostics=True,
                        generator=torch.Generator(device=cuda_device).manual_seed(dim))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    plain = s.sample(x=x0.to(cuda_device), n_steps=T, generator=torch.Generator(device=cuda_device).manual_seed(dim))
    assert torch.equal(tr[:, -1], plain)
    torch.testing.assert_close(diag["mean"].double(), tr.double().mean(dim=0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(diag["var"].double(), tr.double().var(dim=0, unbiased=False), rtol=1e-4, atol=1e-7)
