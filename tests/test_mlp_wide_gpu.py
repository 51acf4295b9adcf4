"""The fused MLP energy beyond the two-moons shape (csrc/mlp_wide.hip): hidden width 64 / 128 / 256 (256: weights streamed from L2), input dim up to 128 --
the reference's benchmark network (benchmarks/registry.py:372-387: Linear(dim,128)-SiLU-Linear(128,128)-SiLU-
Linear(128,1) at dim 8 / 32 / 128).  One evaluation against autograd (and against the fp64 network), the k-fused
chain against the CPU autograd chain on injected noise, the sampler's routes, and the native-RNG field."""

import copy

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib, _rng
from torchebm_amd.samplers.langevin import em_coefficients

pytestmark = pytest.mark.gpu

H256 = 256 in ta.MLPEnergy.FUSED_HIDDEN  # the streamed-weight family is not in the shipped library since round 5 (make H256=1)


def _built(shapes):
    """The parametrisation without the hidden-256 shapes unless this build has their kernels."""
    return [s for s in shapes if H256 or 256 not in s[1:2]]



def _models(cuda_device, in_dim, hidden, seed=0, scale=1.0):
    torch.manual_seed(seed)
    cpu = ta.MLPEnergy(in_dim, hidden)
    with torch.no_grad():
        for p in cpu.parameters():
            p.mul_(scale)
    return cpu, copy.deepcopy(cpu).to(cuda_device)


@pytest.mark.parametrize("in_dim,hidden", _built([(8, 128), (32, 128), (128, 128), (5, 128), (33, 128), (100, 128), (64, 64), (7, 64),
                                           (128, 64), (2, 64), (32, 256), (8, 256), (33, 256), (64, 256), (100, 256), (128, 256)]))
def test_energy_and_gradient_match_autograd(cuda_device, in_dim, hidden):
    cpu, gpu = _models(cuda_device, in_dim, hidden, seed=in_dim + hidden, scale=1.5)
    spec = gpu.fused_spec()
    assert spec is not None and spec.dim == in_dim and spec.hmc is (in_dim <= ta.MLPEnergy.HMC_MAX_DIM[hidden])
    n = 333  # not a multiple of the 32-chain tile
    x = torch.randn(n, in_dim, generator=torch.Generator().manual_seed(1)) * 1.5
    x_d = x.to(cuda_device)
    e, g = torch.empty(n, device=cuda_device), torch.empty(n, in_dim, device=cuda_device)
    _lib.call("ebm_energy_grad_f32", spec.to_c(), x_d.data_ptr(), n, in_dim, e.data_ptr(), g.data_ptr(), _lib.stream_handle(cuda_device))
    want_e, want_g = cpu(x), cpu.gradient(x)
    # both are fp32 evaluations in different summation orders: compare with the fp64 network as the referee
    cpu64 = copy.deepcopy(cpu).double()
    x64 = x.double().requires_grad_(True)
    e64 = cpu64(x64)
    (g64,) = torch.autograd.grad(e64.sum(), x64)
    err_hip = (g.cpu().double() - g64).abs().max().item()
    err_torch = (want_g.double() - g64).abs().max().item()
    scale = g64.abs().max().item()
    assert err_hip <= max(4.0 * err_torch, 2e-6 * scale), (err_hip, err_torch, scale)
    torch.testing.assert_close(e.cpu(), want_e.detach(), rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(g.cpu(), want_g, rtol=2e-4, atol=2e-5 * max(scale, 1.0))


@pytest.mark.parametrize("in_dim,hidden", _built([(8, 128), (32, 128), (128, 128), (30, 64), (32, 256), (30, 256), (128, 256)]))
def test_fused_chain_matches_cpu_autograd_chain_with_injected_noise(cuda_device, in_dim, hidden):
    """The k-fused chain against the CPU autograd chain on the same noise, refereed by the fp64 network's chain: both
    fp32 chains drift from it at the rate the dynamics amplify round-off, and the kernel may not drift faster than
    4 x what torch's own fp32 arithmetic does, at any kept step (no flat tolerance)."""
    cpu, gpu = _models(cuda_device, in_dim, hidden, seed=5)
    cpu64 = copy.deepcopy(cpu).double()
    n, k, eta, sigma = 200, 12, 0.05, 0.7
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(n, in_dim, generator=g)
    noise = torch.randn(k, n, in_dim, generator=g)
    want, ref = x0, x0.double()
    rows, rows64 = [], []
    for i in range(k):
        want = oracle.em_step(want, cpu.gradient(want), noise[i], eta, sigma)
        x64 = ref.clone().requires_grad_(True)
        (g64,) = torch.autograd.grad(cpu64(x64).sum(), x64)
        ref = oracle.em_step(ref, g64, noise[i].double(), eta, sigma)
        if (i + 1) % 4 == 0:
            rows.append(want)
            rows64.append(ref)
    x = x0.to(cuda_device)
    a, sq, coef = em_coefficients(eta, sigma)
    traj = torch.empty(n, k // 4, in_dim, device=cuda_device)
    nz = noise.to(cuda_device)
    _lib.call("ebm_langevin_chain_f32", gpu.fused_spec().to_c(), x.data_ptr(), n, in_dim, k, a, sq, coef, None, 0, 0.0, 0.0, 4,
              traj.data_ptr(), None, nz.data_ptr(), 0, 0, _lib.stream_handle(cuda_device))
    got = traj.cpu()
    assert torch.equal(got[:, -1], x.cpu())
    for j in range(k // 4):
        scale = rows64[j].abs().max().item()
        err_hip = (got[:, j].double() - rows64[j]).abs().max().item()
        err_torch = (rows[j].double() - rows64[j]).abs().max().item()
        assert err_hip <= max(4.0 * err_torch, 4e-6 * scale), (j, err_hip, err_torch, scale)


def test_sampler_routes_and_native_rng_field(cuda_device):
    """LangevinDynamics on the benchmark MLP is ONE launch; its Philox field is the (seed, step, element) field of
    every other kernel, so the step route (autograd gradient + ebm_langevin_step_f32) lands on the same chains."""

    class Sub(ta.MLPEnergy):
        def forward(self, x):
            return super().forward(x)

    for in_dim in (8, 30, 128):
        torch.manual_seed(in_dim)
        fused_model = ta.MLPEnergy(in_dim, 128, device=cuda_device)
        step_model = Sub(in_dim, 128, device=cuda_device)
        step_model.load_state_dict(fused_model.state_dict())
        assert step_model.fused_spec() is None
        x0 = torch.randn(1000, in_dim, device=cuda_device)
        sf = ta.LangevinDynamics(fused_model, step_size=0.05, clamp=(-3.0, 3.0), device=cuda_device)
        ss = ta.LangevinDynamics(step_model, step_size=0.05, clamp=(-3.0, 3.0), device=cuda_device)
        c0 = hip_calls("ebm_langevin_chain_f32")
        a = sf.sample(x=x0, n_steps=10, generator=torch.Generator(device=cuda_device).manual_seed(4))
        b = ss.sample(x=x0, n_steps=10, generator=torch.Generator(device=cuda_device).manual_seed(4))
        assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-3)
        # trajectory + diagnostics (statistics from the state between launches for this energy)
        traj, diag = sf.sample(x=x0, n_steps=6, thin=2, return_trajectory=True, return_diagnostics=True)
        assert traj.shape == (1000, 3, in_dim) and torch.isfinite(traj).all()
        torch.testing.assert_close(diag["energy"][-1], fused_model(traj[:, -1]).mean(), rtol=1e-4, atol=1e-4)
    # every shape the Langevin kernels take has an HMC transition kernel too; other hidden widths are refused by the C ABI
    from torchebm_amd.core.energies import FusedSpec

    odd = FusedSpec(_lib.ENERGY_MLP, n_comp=96, dev0=torch.zeros(96 * 8 + 96 + 96 * 96 + 96 + 96 + 1, device=cuda_device), langevin_only=True, dim=8)
    xx = torch.zeros(4, 8, device=cuda_device)
    with pytest.raises((RuntimeError, ValueError), match="hidden width"):
        _lib.call("ebm_hmc_chain_f32", odd.to_c(), xx.data_ptr(), 4, 8, 1, 1, 0.01, None, 0, 0.0, None, 1, None, None, None, None,
                  None, None, 0, 0, _lib.stream_handle(cuda_device))
    # a hidden width outside 64 / 128 / 256 through the sampler: the per-transition route on autograd
    h = ta.HamiltonianMonteCarlo(ta.MLPEnergy(8, 96, device=cuda_device), step_size=0.05, n_leapfrog_steps=3, device=cuda_device)
    c0 = hip_calls("ebm_hmc_chain_f32")
    out = h.sample(x=torch.randn(64, 8, device=cuda_device), n_steps=2)
    assert hip_calls("ebm_hmc_chain_f32") == c0 and torch.isfinite(out).all()


class _CpuMlpEnergy:
    def __init__(self, model):
        self.model = model

    def energy(self, x):
        return self.model(x).detach()

    def grad(self, x):
        return self.model.gradient(x)


@pytest.mark.parametrize("in_dim,hidden,mass", _built([(8, 128, None), (32, 128, 1.7), (30, 128, "diag"), (64, 128, None), (33, 64, "diag"),
                                                (64, 64, 0.6), (5, 64, None), (32, 256, None), (17, 256, "diag"), (8, 256, 2.0),
                                                (96, 128, None), (100, 128, "diag"), (128, 128, 1.3), (128, 64, "diag"), (90, 64, None),
                                                (64, 256, None), (50, 256, "diag"), (96, 256, None), (100, 256, "diag"), (128, 256, 0.8)]))
def test_wide_hmc_kernel_matches_cpu_autograd_chain_with_injected_noise(cuda_device, in_dim, hidden, mass):
    """csrc/mlp_wide_hmc.hip through the C ABI against the oracle's HMC on the CPU autograd network: same momenta, same
    uniforms, every mass form, thinning; accept decisions identical except within fp32 round-off of u."""
    cpu, gpu = _models(cuda_device, in_dim, hidden, seed=10 + in_dim, scale=1.2)
    n, T, L, eps = 515, 4, 5, 0.05
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(n, in_dim, generator=g)
    p = torch.randn(T, n, in_dim, generator=g)
    u = torch.rand(T, n, generator=g)
    if mass == "diag":
        mass = torch.rand(in_dim, generator=g) + 0.5
    want = oracle.hmc_chain(_CpuMlpEnergy(cpu), x0, p, u, [eps] * T, L, mass=mass, thin=2, want_traj=True)
    # the fp64 referee: the same chain on the fp64 network; a decision whose margin |u - a| exceeds 1e-4 there is out of
    # reach of fp32 round-off, so a chain all of whose decisions are that clear must be decided identically by the kernel
    ref = oracle.hmc_chain(_CpuMlpEnergy(copy.deepcopy(cpu).double()), x0.double(), p.double(), u.double(), [eps] * T, L,
                           mass=mass.double() if torch.is_tensor(mass) else mass, thin=2, want_traj=True, want_margins=True)
    from torchebm_amd.integrators.symplectic import _mass_args

    spec = gpu.fused_spec()
    assert spec.hmc
    x = x0.to(cuda_device).clone()
    kind, m_scalar, m_diag = _mass_args(mass.to(cuda_device) if torch.is_tensor(mass) else mass, x)
    traj = torch.empty(n, T // 2, in_dim, device=cuda_device)
    mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
    counts = torch.zeros(T, dtype=torch.int32, device=cuda_device)
    p_d, u_d = p.to(cuda_device).contiguous(), u.to(cuda_device).contiguous()
    _lib.call("ebm_hmc_chain_f32", spec.to_c(), x.data_ptr(), n, in_dim, T, L, eps, None, kind, m_scalar, _lib.ptr(m_diag), 2,
              traj.data_ptr(), None, mask.data_ptr(), counts.data_ptr(), p_d.data_ptr(), u_d.data_ptr(), 0, 0,
              _lib.stream_handle(cuda_device))
    got_mask = mask.cpu().bool()
    agree = (got_mask == want["accepted"]).all(dim=0)
    assert agree.float().mean().item() >= 0.99
    clear = ref["margins"].min(dim=0).values > 1e-4
    assert clear.float().mean().item() > 0.9 and (got_mask[:, clear] == ref["accepted"][:, clear]).all()
    assert torch.equal(counts.cpu().long(), got_mask.sum(dim=1))
    err = ((traj.cpu() - want["trajectory"]).abs() / want["trajectory"].abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
    assert (err[agree] <= 2e-3).all()
    # positions of the clearly decided chains: the kernel may not be further from the fp64 chain than 4 x torch's fp32 one
    err_hip = (traj.cpu().double() - ref["trajectory"])[clear].abs().max().item()
    err_torch = (want["trajectory"].double() - ref["trajectory"])[clear].abs().max().item()
    assert err_hip <= max(4.0 * err_torch, 4e-6 * ref["trajectory"].abs().max().item()), (err_hip, err_torch)
    assert torch.equal(traj[:, -1], x)


@pytest.mark.parametrize("in_dim,hidden", _built([(16, 128), (48, 64), (32, 256), (128, 128), (100, 64), (64, 256), (128, 256)]))
def test_sampler_hmc_on_the_wide_mlp_is_one_launch_on_the_shared_field(cuda_device, in_dim, hidden):
    class Sub(ta.MLPEnergy):
        def forward(self, x):
            return super().forward(x)

    torch.manual_seed(4)
    fused_model = ta.MLPEnergy(in_dim, hidden, device=cuda_device)
    step_model = Sub(in_dim, hidden, device=cuda_device)
    step_model.load_state_dict(fused_model.state_dict())
    x0 = torch.randn(3000, in_dim, device=cuda_device)
    kw = dict(step_size=0.05, n_leapfrog_steps=5, device=cuda_device)
    hf, hs = ta.HamiltonianMonteCarlo(fused_model, **kw), ta.HamiltonianMonteCarlo(step_model, **kw)
    c0 = hip_calls("ebm_hmc_chain_f32")
    a = hf.sample(x=x0, n_steps=6, generator=torch.Generator(device=cuda_device).manual_seed(8))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    b = hs.sample(x=x0, n_steps=6, generator=torch.Generator(device=cuda_device).manual_seed(8))
    same = ((a - b).abs().amax(dim=1) <= 5e-3).float().mean().item()
    assert same >= 0.99, same
    _, da = hf.sample(x=x0, n_steps=4, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(9))
    _, db = hs.sample(x=x0, n_steps=4, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(9))
    torch.testing.assert_close(da["acceptance_rate"], db["acceptance_rate"], rtol=0, atol=3e-3)
    torch.testing.assert_close(da["energy"], db["energy"], rtol=1e-3, atol=1e-3)


def test_wide_hmc_safe_mode_on_extreme_states(cuda_device):
    torch.manual_seed(1)
    model = ta.MLPEnergy(24, 128, device=cuda_device)
    x0 = torch.randn(200, 24, device=cuda_device)
    x0[5] = 1e30
    x0[70, 3] = float("inf")
    x0[131, 20] = float("nan")
    h = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=4, device=cuda_device)
    out = h.sample(x=x0, n_steps=3, generator=torch.Generator(device=cuda_device).manual_seed(2))
    ok = torch.ones(200, dtype=torch.bool, device=cuda_device)
    ok[[5, 70, 131]] = False
    assert torch.isfinite(out[ok]).all()
    # the poisoned chains do not leak into their wave: the clean chains land where they land without them
    clean = h.sample(x=x0[ok], n_steps=3, generator=torch.Generator(device=cuda_device).manual_seed(2))
    assert clean.shape == (197, 24)


@pytest.mark.skipif(not H256, reason="hidden width 256 is not in this build (make H256=1)")
@pytest.mark.parametrize("in_dim", [32, 30, 128])
def test_hidden_256_sampler_is_one_launch_on_the_shared_field(cuda_device, in_dim):
    """Hidden width 256 (the streamed-weight variant): ``LangevinDynamics`` is one ``ebm_langevin_chain_f32`` launch, its
    noise is the (seed, step, element) field of the step route, thinning / clamp / a scheduled step size included."""

    class Sub(ta.MLPEnergy):
        def forward(self, x):
            return super().forward(x)

    torch.manual_seed(in_dim)
    fused_model = ta.MLPEnergy(in_dim, 256, device=cuda_device)
    step_model = Sub(in_dim, 256, device=cuda_device)
    step_model.load_state_dict(fused_model.state_dict())
    assert fused_model.fused_spec() is not None and step_model.fused_spec() is None
    x0 = torch.randn(777, in_dim, device=cuda_device)
    from torchebm_amd.core import LinearScheduler

    sf = ta.LangevinDynamics(fused_model, step_size=LinearScheduler(0.05, 0.02, 8), clamp=(-3.0, 3.0), device=cuda_device)
    ss = ta.LangevinDynamics(step_model, step_size=LinearScheduler(0.05, 0.02, 8), clamp=(-3.0, 3.0), device=cuda_device)
    c0 = hip_calls("ebm_langevin_chain_f32")
    a = sf.sample(x=x0, n_steps=8, thin=2, return_trajectory=True, generator=torch.Generator(device=cuda_device).manual_seed(9))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
    b = ss.sample(x=x0, n_steps=8, thin=2, return_trajectory=True, generator=torch.Generator(device=cuda_device).manual_seed(9))
    assert a.shape == (777, 4, in_dim)
    torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-3)


@pytest.mark.parametrize("in_dim,hidden", _built([(8, 128), (32, 256), (100, 64)]))
def test_gradient_is_one_hip_launch(cuda_device, in_dim, hidden):
    """``MLPEnergy.gradient`` on a CUDA fp32 state is one ``ebm_energy_grad_f32`` launch (forward + input gradient on the
    matrix cores), equal to autograd within fp32 summation order; conditioning kwargs, a subclass with its own
    ``forward``, a 3-D state or a CPU model stay on autograd."""
    cpu, gpu = _models(cuda_device, in_dim, hidden, seed=11, scale=1.5)
    x = torch.randn(501, in_dim, generator=torch.Generator().manual_seed(2))
    c0 = hip_calls("ebm_energy_grad_f32")
    got = gpu.gradient(x.to(cuda_device))
    assert hip_calls("ebm_energy_grad_f32") == c0 + 1
    want = cpu.gradient(x)
    assert got.shape == want.shape and not got.requires_grad
    torch.testing.assert_close(got.cpu(), want, rtol=2e-4, atol=2e-5 * max(want.abs().max().item(), 1.0))

    class Sub(ta.MLPEnergy):
        def forward(self, x):
            return super().forward(x)

    sub = Sub(in_dim, hidden, device=cuda_device)
    sub.load_state_dict(gpu.state_dict())
    c0 = hip_calls("ebm_energy_grad_f32")
    torch.testing.assert_close(sub.gradient(x.to(cuda_device)), got, rtol=2e-4, atol=2e-4)
    cpu.gradient(x)
    assert hip_calls("ebm_energy_grad_f32") == c0
    # parameter gradients are untouched: forward() is autograd
    e = gpu(x.to(cuda_device)).sum()
    e.backward()
    assert all(p.grad is not None for p in gpu.parameters())


def test_hmc_on_the_wide_mlp_uses_the_hip_gradient(cuda_device):
    """Forced onto the per-transition route (as a non-fusable mass or integrator would), HMC on an MLP energy (HIP kicks around ``model.gradient``) now
    evaluates each of its L + 1 forces in one launch.  Same momenta, same uniforms as the autograd route."""

    class Sub(ta.MLPEnergy):
        def forward(self, x):
            return super().forward(x)

    torch.manual_seed(3)
    fast = ta.MLPEnergy(80, 128, device=cuda_device)
    slow = Sub(80, 128, device=cuda_device)
    slow.load_state_dict(fast.state_dict())
    x0 = torch.randn(2000, 80, device=cuda_device)
    outs = []
    for model in (fast, slow):
        h = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=4, device=cuda_device)
        h.capture_graph = False
        h._route = lambda x_, kw_: ("step", None)  # the route a non-fusable configuration takes
        c0 = hip_calls("ebm_energy_grad_f32")
        out, diag = h.sample(x=x0, n_steps=3, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(5))
        outs.append((out, diag, hip_calls("ebm_energy_grad_f32") - c0))
    assert outs[0][2] >= 3 * 4 and outs[1][2] == 0
    close = ((outs[0][0] - outs[1][0]).abs().amax(dim=1) <= 2e-3).float().mean().item()
    assert close >= 0.995, close  # a Metropolis tie may flip on a summation-order difference
    torch.testing.assert_close(outs[0][1]["acceptance_rate"], outs[1][1]["acceptance_rate"], rtol=0, atol=5e-3)


@pytest.mark.skipif(not H256, reason="hidden width 256 is not in this build (make H256=1)")
def test_streamed_weights_must_be_16_byte_aligned(cuda_device):
    """H = 256 reads its weights with 16-byte loads straight from the parameter block: a misaligned block is refused by
    the C ABI (EBM_EINVAL -> ValueError), never read."""
    from torchebm_amd.core.energies import FusedSpec

    model = ta.MLPEnergy(32, 256, device=cuda_device)
    good = model.fused_spec()
    shifted = torch.empty(good.dev0.numel() + 1, device=cuda_device)
    shifted[1:] = good.dev0
    bad = FusedSpec(_lib.ENERGY_MLP, n_comp=256, dev0=shifted[1:], langevin_only=True, dim=32, hmc=True)
    x = torch.zeros(64, 32, device=cuda_device)
    with pytest.raises((ValueError, RuntimeError), match="16-byte aligned"):
        _lib.call("ebm_langevin_chain_f32", bad.to_c(), x.data_ptr(), 64, 32, 2, 0.01, 0.1, 1.0, None, 0, 0.0, 0.0, 1, None, None, None, 0, 0,
                  _lib.stream_handle(cuda_device))
    with pytest.raises((ValueError, RuntimeError), match="16-byte aligned"):
        _lib.call("ebm_hmc_chain_f32", bad.to_c(), x.data_ptr(), 64, 32, 1, 1, 0.01, None, 0, 0.0, None, 1, None, None, None, None,
                  None, None, 0, 0, _lib.stream_handle(cuda_device))
    assert torch.equal(x, torch.zeros_like(x))


@pytest.mark.parametrize("in_dim,hidden", _built([(20, 128), (70, 64), (40, 256)]))
def test_wide_hmc_edge_shapes_schedule_and_prefix_identity(cuda_device, in_dim, hidden):
    """One chain, a batch that is not a multiple of the 32-chain tile, a scheduled step size with thinning that does not
    divide the transitions, and prefix identity: a chain's native-RNG trajectory depends on its index and the seed only,
    not on how many other chains share the launch."""
    from torchebm_amd.core import LinearScheduler

    torch.manual_seed(7)
    model = ta.MLPEnergy(in_dim, hidden, device=cuda_device)
    x0 = torch.randn(161, in_dim, device=cuda_device)

    def run(x, **kw):
        h = ta.HamiltonianMonteCarlo(model, step_size=LinearScheduler(0.08, 0.03, 5), n_leapfrog_steps=3, device=cuda_device)
        c0 = hip_calls("ebm_hmc_chain_f32")
        out = h.sample(x=x, n_steps=5, generator=torch.Generator(device=cuda_device).manual_seed(21), **kw)
        assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
        return out

    full = run(x0, thin=2, return_trajectory=True)
    assert full.shape == (161, 2, in_dim) and torch.isfinite(full).all()
    again = run(x0, thin=2, return_trajectory=True)
    assert torch.equal(full, again)                                   # deterministic
    for n in (1, 33, 128):
        part = run(x0[:n].clone(), thin=2, return_trajectory=True)
        assert torch.equal(part, full[:n])                            # prefix identity
    last = run(x0)
    assert last.shape == (161, in_dim) and not torch.equal(last, x0)
    # the kept states are transitions 2 and 4; the final state (after 5) differs from the last kept one for most chains
    assert (last != full[:, -1]).any(dim=1).float().mean().item() > 0.5
    assert torch.equal(x0, x0.clone())                                # the caller's tensor is not the in/out buffer


@pytest.mark.parametrize("in_dim,hidden,n", _built([(2, 128, 4099), (32, 128, 1000), (64, 128, 515), (33, 64, 777), (32, 256, 300), (100, 128, 257)]))
def test_langevin_diagnostics_come_from_records_of_the_one_chain_launch(cuda_device, in_dim, hidden, n):
    """return_diagnostics=True on the MLP energy stays ONE chain launch -- every wave stores the record
    of its 32 chains at the kept steps (the energy share one evaluation later), ebm_diag_finish_f32 merges them -- and
    equals the torch reductions of the stored trajectory (langevin_dynamics.py:170-185)."""
    torch.manual_seed(in_dim)
    model = ta.MLPEnergy(in_dim, hidden, device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.05, device=cuda_device)
    x0 = torch.randn(n, in_dim, device=cuda_device)
    for k, thin in ((12, 3), (10, 4), (5, 1)):   # kept last step; unkept tail; every step
        c0, f0 = hip_calls("ebm_langevin_chain_f32"), hip_calls("ebm_diag_finish_f32")
        traj, diag = s.sample(x=x0, n_steps=k, thin=thin, return_trajectory=True, return_diagnostics=True,
                              generator=torch.Generator(device=cuda_device).manual_seed(5))
        assert hip_calls("ebm_langevin_chain_f32") == c0 + 1 and hip_calls("ebm_diag_finish_f32") == f0 + 1
        plain = s.sample(x=x0, n_steps=k, thin=thin, return_trajectory=True, generator=torch.Generator(device=cuda_device).manual_seed(5))
        assert torch.equal(traj, plain)                      # the records do not touch the chains
        t64 = traj.double()
        torch.testing.assert_close(diag["mean"].double(), t64.mean(dim=0), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(diag["var"].double(), t64.var(dim=0, unbiased=False).clamp(1e-10, 1e10), rtol=1e-4, atol=1e-7)
        want_e = torch.stack([model(traj[:, j]).double().mean() for j in range(traj.shape[1])])
        torch.testing.assert_close(diag["energy"].double(), want_e, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("in_dim,hidden,mass", _built([(2, 128, None), (32, 128, 1.7), (48, 64, "diag"), (32, 256, None)]))
def test_hmc_diagnostics_come_from_records_of_the_one_chain_launch(cuda_device, in_dim, hidden, mass):
    torch.manual_seed(3 + in_dim)
    model = ta.MLPEnergy(in_dim, hidden, device=cuda_device)
    if mass == "diag":
        mass = torch.rand(in_dim, device=cuda_device) + 0.5
    h = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=4, mass=mass, device=cuda_device)
    n = 1234
    x0 = torch.randn(n, in_dim, device=cuda_device)
    c0 = hip_calls("ebm_hmc_chain_f32")
    traj, diag = h.sample(x=x0, n_steps=6, thin=1, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(2))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    t64 = traj.double()
    torch.testing.assert_close(diag["mean"].double(), t64.mean(dim=0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(diag["var"].double(), t64.var(dim=0, unbiased=False).clamp(1e-10, 1e10), rtol=1e-4, atol=1e-7)
    want_e = torch.stack([model(traj[:, j]).double().clamp(-1e10, 1e10).mean() for j in range(6)])
    torch.testing.assert_close(diag["energy"].double(), want_e, rtol=1e-4, atol=1e-5)
    prev = torch.cat([x0.unsqueeze(1), traj[:, :-1]], dim=1)
    moved = (traj != prev).any(dim=2).double().mean(dim=0)     # an accepted proposal moves the chain
    torch.testing.assert_close(diag["acceptance_rate"].double(), moved, rtol=0, atol=1e-6)
    assert 0.2 < float(diag["acceptance_rate"].mean()) <= 1.0


@pytest.mark.parametrize("hidden", [64, 128])
def test_thin_input_kernels_track_the_matrix_pipe_kernels(cuda_device, hidden):
    """dim <= 2 (config 5's shape): the plain call runs the MODE 4 kernels -- W1's two contractions on the vector unit in exact fp32,
    csrc/mlp_wide_thin.hip -- while a call with an (inactive) clamp takes the general MODE 2 kernel on the same noise field (same
    generator seed): two fp32-accurate evaluations of the same chain.  Bar: after k steps they differ by no more than fp32 chain
    drift (the fp64 chain is the referee in the tests above; here 2e-4 of the state's scale at k = 15), and the energies / gradients
    of the two agree to 1e-5 relative."""
    cpu, gpu = _models(cuda_device, 2, hidden, seed=9, scale=1.3)
    n, k = 4099, 15
    x0 = torch.randn(n, 2, device=cuda_device)
    plain = ta.LangevinDynamics(gpu, step_size=0.02, noise_scale=1.0, device=cuda_device)
    clamped = ta.LangevinDynamics(gpu, step_size=0.02, noise_scale=1.0, clamp=(-1e30, 1e30), device=cuda_device)
    a = plain.sample(x=x0, n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(5))
    b = clamped.sample(x=x0, n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(5))
    assert not torch.equal(a, x0) and torch.isfinite(a).all()
    scale = b.abs().max().item()
    assert (a - b).abs().max().item() <= 2e-4 * scale, ((a - b).abs().max().item(), scale)
    # the training forward (ebm_mlp_backward_acts_f32 on the thin kernel) against autograd
    e = gpu(x0)
    assert type(e.grad_fn).__name__.startswith("_FusedMLPTraining")
    ref = gpu.net(x0).squeeze(-1)
    assert ((e - ref).abs() / (1 + ref.abs())).max().item() <= 2e-5


@pytest.mark.parametrize("hidden", [64, 128])
def test_thin_chain_kernel_against_the_fp64_chain_on_its_own_noise(cuda_device, hidden):
    """The MODE 4 chain kernel (dim 2, native Philox noise -- the plain call) with the fp64 network's chain as referee: the noise
    field of the launch is materialised with ebm_noise_fill_f32 (same seed, same steps) and fed to the CPU chains; the kernel may
    not drift from the fp64 chain faster than 4 x what torch's own fp32 arithmetic does (the bar of the injected-noise tests above,
    which run the general kernel)."""
    from torchebm_amd.samplers.langevin import em_coefficients

    cpu, gpu = _models(cuda_device, 2, hidden, seed=21, scale=1.2)
    cpu64 = copy.deepcopy(cpu).double()
    n, k, eta, sigma = 2000, 12, 0.05, 0.7
    x0 = torch.randn(n, 2, generator=torch.Generator().manual_seed(4))
    seed, step0 = 0x5EED1234, 77
    a, sq, coef = em_coefficients(eta, sigma)
    x = x0.to(cuda_device).clone()
    spec = gpu.fused_spec()
    c0 = hip_calls("ebm_langevin_chain_f32")
    _lib.call("ebm_langevin_chain_f32", spec.to_c(), x.data_ptr(), n, 2, k, a, sq, coef, None, 0, 0.0, 0.0, 1, None, None, None, seed, step0,
              _lib.stream_handle(cuda_device))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
    noise = torch.empty(k, n, 2, device=cuda_device)
    for i in range(k):
        _lib.call("ebm_noise_fill_f32", noise[i].data_ptr(), n * 2, _lib.NOISE_NORMAL, seed, step0 + i, _lib.stream_handle(cuda_device))
    noise = noise.cpu()
    x32, x64 = x0.clone(), x0.double()
    for i in range(k):  # the reference's update order (base_integrator.py:673-731) in fp32 and in fp64
        x32 = (x32 - eta * cpu.gradient(x32)) + coef * (noise[i] * sq)
        x64 = (x64 - eta * cpu64.gradient(x64)) + coef * (noise[i].double() * sq)
    err_hip = (x.cpu().double() - x64).abs()
    err_ref = (x32.double() - x64).abs()
    assert err_hip.max().item() <= 4 * err_ref.max().item() + 1e-6, (err_hip.max().item(), err_ref.max().item())
    assert err_hip.median().item() <= 4 * err_ref.median().item() + 1e-7
