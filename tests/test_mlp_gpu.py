"""the fused MLP-energy Langevin kernel (forward + input-gradient on the matrix
cores, weights in LDS) against autograd.  Tolerances: fp32 MFMA is an exact fmaf chain, but rocBLAS /
CPU GEMMs sum in another order and the kernel uses the hardware exp/rcp for the sigmoid:
|dE| <= 2e-5 * (1 + |E|), |dg| <= 2e-4 * (1 + |g|) for one evaluation; a k-step chain amplifies
that by the dynamics."""

import copy

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients
from torchebm_amd.utils.synthetic import two_moons

pytestmark = pytest.mark.gpu


def _models(cuda_device, in_dim=2, seed=0, scale=1.0):
    torch.manual_seed(seed)
    cpu = ta.MLPEnergy(in_dim)
    with torch.no_grad():
        for p in cpu.parameters():
            p.mul_(scale)
    return cpu, copy.deepcopy(cpu).to(cuda_device)


@pytest.mark.parametrize("in_dim,n", [(2, 1000), (1, 77), (3, 256), (4, 129)])
def test_energy_and_gradient_match_autograd(cuda_device, in_dim, n):
    cpu, gpu = _models(cuda_device, in_dim, seed=in_dim, scale=2.0)
    spec = gpu.fused_spec()
    assert spec is not None and spec.kind == _lib.ENERGY_MLP and spec.langevin_only
    x = torch.randn(n, in_dim) * 2
    e = torch.empty(n, device=cuda_device)
    g = torch.empty(n, in_dim, device=cuda_device)
    x_d = x.to(cuda_device)  # keep the device copy alive across the launch
    _lib.call("ebm_energy_grad_f32", spec.to_c(), x_d.data_ptr(), n, in_dim, e.data_ptr(), g.data_ptr(),
              _lib.stream_handle(cuda_device))
    want_e = cpu(x.double().float()).detach()
    want_g = cpu.gradient(x)
    # fp64 reference of the same network, to show both fp32 paths are equally close to the truth
    ref = copy.deepcopy(cpu).double()
    e64 = ref(x.double()).detach()
    assert ((e.cpu() - want_e).abs() / (1 + want_e.abs())).max().item() <= 2e-5
    assert ((g.cpu() - want_g).abs() / (1 + want_g.abs())).max().item() <= 2e-4
    assert ((e.cpu().double() - e64).abs() / (1 + e64.abs())).max().item() <= 2e-5


def test_fused_chain_matches_cpu_autograd_chain_with_injected_noise(cuda_device):
    cpu, gpu = _models(cuda_device, 2, seed=5)
    n, k, eta, sigma = 513, 12, 0.05, 1.0
    x0 = two_moons(n, 0.05, seed=3)
    noise = torch.randn(k, n, 2, generator=torch.Generator().manual_seed(9))
    want = x0.clone()
    for i in range(k):
        want = oracle.em_step(want, cpu.gradient(want), noise[i], eta, sigma)
    spec = gpu.fused_spec()
    x = x0.to(cuda_device).clone()
    a, sq, coef = em_coefficients(eta, sigma)
    traj = torch.empty(n, k // 4, 2, device=cuda_device)
    noise_d = noise.to(cuda_device)
    _lib.call("ebm_langevin_chain_f32", spec.to_c(), x.data_ptr(), n, 2, k, a, sq, coef, None, 0, 0.0, 0.0, 4, traj.data_ptr(),
              None, noise_d.data_ptr(), 0, 0, _lib.stream_handle(cuda_device))
    torch.testing.assert_close(x.cpu(), want, rtol=1e-3, atol=1e-3)
    assert torch.equal(traj[:, -1], x)


def test_sampler_fused_route_equals_step_route_noise_field(cuda_device):
    """LangevinDynamics on MLPEnergy takes the fused route (one launch); a subclass takes the
    per-step route (autograd + update kernel).  Same generator => same Philox field => the two
    agree to gradient round-off."""

    class Sub(ta.MLPEnergy):
        def forward(self, x):
            return super().forward(x)

    torch.manual_seed(2)
    fused_model = ta.MLPEnergy(2, device=cuda_device)
    step_model = Sub(2, device=cuda_device)
    step_model.load_state_dict(fused_model.state_dict())
    assert step_model.fused_spec() is None
    x0 = two_moons(4096, 0.05, seed=1, device=cuda_device)
    sf = ta.LangevinDynamics(fused_model, step_size=0.1, clamp=(-3.0, 3.0), device=cuda_device)
    ss = ta.LangevinDynamics(step_model, step_size=0.1, clamp=(-3.0, 3.0), device=cuda_device)
    ss.capture_graph = False  # count the eager launches of the step route
    c0, s0 = hip_calls("ebm_langevin_chain_f32"), hip_calls("ebm_langevin_step_f32")
    a = sf.sample(x=x0, n_steps=20, generator=torch.Generator(device=cuda_device).manual_seed(4))
    b = ss.sample(x=x0, n_steps=20, generator=torch.Generator(device=cuda_device).manual_seed(4))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 1 and hip_calls("ebm_langevin_step_f32") == s0 + 20
    torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-3)
    assert a.abs().max().item() <= 3.0
    # diagnostics / trajectory path
    traj, diag = sf.sample(x=x0, n_steps=6, thin=2, return_trajectory=True, return_diagnostics=True)
    assert traj.shape == (4096, 3, 2) and torch.isfinite(diag["energy"]).all()
    torch.testing.assert_close(diag["energy"][-1], fused_model(traj[:, -1]).mean(), rtol=1e-4, atol=1e-4)


def test_pcd_training_with_fused_mlp_sampler(cuda_device):
    """Config 5's training loop with the fused sampler: weights are re-read at every sample() call,
    so the optimiser's in-place updates are seen by the next launch."""
    torch.manual_seed(0)
    model = ta.MLPEnergy(2, device=cuda_device)
    n, k = 65536, 20
    data = two_moons(n, 0.05, seed=0, device=cuda_device)
    sampler = ta.LangevinDynamics(model, step_size=0.1, noise_scale=1.0, device=cuda_device)
    pcd = ta.ContrastiveDivergence(model, sampler, k_steps=k, persistent=True, buffer_size=n, init_steps=0, device=cuda_device)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    c0 = hip_calls("ebm_langevin_chain_f32")
    gaps = []
    for _ in range(4):
        loss, neg = pcd(data)
        opt.zero_grad()
        loss.backward()
        opt.step()
        gaps.append((model(neg).mean() - model(data).mean()).item())
        assert torch.isfinite(loss) and torch.isfinite(neg).all()
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 4
    # the fused kernel saw the updated weights: same start + same seed, different parameters -> different negatives
    g = torch.Generator(device=cuda_device).manual_seed(1)
    before = sampler.sample(x=data[:1024], n_steps=5, generator=g)
    with torch.no_grad():
        model.net[4].weight.mul_(3.0)
    after = sampler.sample(x=data[:1024], n_steps=5, generator=torch.Generator(device=cuda_device).manual_seed(1))
    assert not torch.equal(before, after)


# ------------------------------------------------------------------------------------------
# HMC on the packaged MLP energy: one launch, the evaluation block inside a transition state machine
# ------------------------------------------------------------------------------------------
class _CpuMlpEnergy:
    """Oracle adapter: the CPU autograd network as the energy object oracle.hmc_chain expects."""

    def __init__(self, model):
        self.model = model

    def energy(self, x):
        return self.model(x).detach()

    def grad(self, x):
        return self.model.gradient(x)


@pytest.mark.parametrize("in_dim,mass", [(2, None), (2, 1.7), (3, "diag"), (4, None), (1, None)])
def test_fused_mlp_hmc_matches_cpu_autograd_chain_with_injected_noise(cuda_device, in_dim, mass):
    cpu, gpu = _models(cuda_device, in_dim, seed=10 + in_dim, scale=1.5)
    n, T, L, eps = 515, 4, 6, 0.07
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(n, in_dim, generator=g)
    p = torch.randn(T, n, in_dim, generator=g)
    u = torch.rand(T, n, generator=g)
    if mass == "diag":
        mass = torch.rand(in_dim, generator=g) + 0.5
    want = oracle.hmc_chain(_CpuMlpEnergy(cpu), x0, p, u, [eps] * T, L, mass=mass, thin=2, want_traj=True)
    from torchebm_amd.integrators.symplectic import _mass_args

    spec = gpu.fused_spec()
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
    agree = (got_mask == want["accepted"]).all(dim=0)           # per chain: every decision identical
    # the kernel's sigmoid uses the hardware exp / rcp: H differs by ~1e-5, so a decision within that of u may flip
    assert agree.float().mean().item() >= 0.99
    assert torch.equal(counts.cpu().long(), got_mask.sum(dim=1))
    err = ((traj.cpu() - want["trajectory"]).abs() / want["trajectory"].abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
    assert (err[agree] <= 2e-3).all()
    assert torch.equal(traj[:, -1], x)


def test_sampler_hmc_on_mlp_energy_is_one_launch_and_tracks_the_step_route(cuda_device):
    class Sub(ta.MLPEnergy):
        def forward(self, x):
            return super().forward(x)

    torch.manual_seed(4)
    fused_model = ta.MLPEnergy(2, device=cuda_device)
    step_model = Sub(2, device=cuda_device)
    step_model.load_state_dict(fused_model.state_dict())
    x0 = two_moons(4096, 0.05, seed=2, device=cuda_device)
    kw = dict(step_size=0.05, n_leapfrog_steps=5, device=cuda_device)
    hf, hs = ta.HamiltonianMonteCarlo(fused_model, **kw), ta.HamiltonianMonteCarlo(step_model, **kw)
    c0 = hip_calls("ebm_hmc_chain_f32")
    a, da = hf.sample(x=x0, n_steps=6, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(8))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1            # diagnostics: records from inside the ONE chain launch
    b, db = hs.sample(x=x0, n_steps=6, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(8))
    # same Philox field: same momenta and uniforms; decisions agree except within round-off of u
    same = ((a - b).abs().amax(dim=1) <= 5e-3).float().mean().item()
    assert same >= 0.99
    torch.testing.assert_close(da["acceptance_rate"], db["acceptance_rate"], rtol=0, atol=2e-3)
    c1 = hip_calls("ebm_hmc_chain_f32")
    out = hf.sample(x=x0, n_steps=10, generator=torch.Generator(device=cuda_device).manual_seed(9))
    assert hip_calls("ebm_hmc_chain_f32") == c1 + 1 and torch.isfinite(out).all()


def test_fused_mlp_hmc_safe_mode_on_extreme_states(cuda_device):
    """Huge / non-finite starts go through the literal path (re-evaluation on the scrubbed position in the
    state machine) without hanging or leaking NaN into other chains of the wave."""
    torch.manual_seed(1)
    model = ta.MLPEnergy(2, device=cuda_device)
    x0 = torch.randn(200, 2, device=cuda_device)
    x0[5] = 1e30
    x0[70] = float("inf")
    x0[131] = float("nan")
    h = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=4, device=cuda_device)
    out = h.sample(x=x0, n_steps=3, generator=torch.Generator(device=cuda_device).manual_seed(2))
    ok = torch.ones(200, dtype=torch.bool, device=cuda_device)
    ok[[5, 70, 131]] = False
    assert torch.isfinite(out[ok]).all()
    clean = h.sample(x=x0[ok], n_steps=3, generator=torch.Generator(device=cuda_device).manual_seed(2))
    assert clean.shape == (197, 2)


def test_adopted_sequential_takes_the_fused_route_and_trains(cuda_device):
    """MLPEnergy.from_sequential shares the user's network: the fused kernel sees the optimiser's updates."""
    from torch import nn

    torch.manual_seed(0)
    net = nn.Sequential(nn.Linear(2, 128), nn.SiLU(), nn.Linear(128, 128), nn.SiLU(), nn.Linear(128, 1)).to(cuda_device)
    energy = ta.MLPEnergy.from_sequential(net)
    assert energy.fused_spec() is not None
    sampler = ta.LangevinDynamics(energy, step_size=0.1, device=cuda_device)
    data = two_moons(2048, 0.05, seed=0, device=cuda_device)
    pcd = ta.ContrastiveDivergence(energy, sampler, k_steps=5, persistent=True, buffer_size=2048, init_steps=0, device=cuda_device)
    opt = torch.optim.Adam(net.parameters(), lr=1e-2)     # optimiser built on the ORIGINAL module
    c0 = hip_calls("ebm_langevin_chain_f32")
    before = sampler.sample(x=data, n_steps=3, generator=torch.Generator(device=cuda_device).manual_seed(1))
    for _ in range(3):
        loss, _ = pcd(data)
        opt.zero_grad()
        loss.backward()
        opt.step()
    after = sampler.sample(x=data, n_steps=3, generator=torch.Generator(device=cuda_device).manual_seed(1))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 5 and not torch.equal(before, after)


# ------------------------------------------------------------------------------------------
# Round 5: the TRAINING forward (parameter gradients only) -- hand-written backward of MLPEnergy.forward
# ------------------------------------------------------------------------------------------
@pytest.mark.parametrize("fused", [False, True])
@pytest.mark.parametrize("in_dim,n", [(2, 65536), (2, 1000), (4, 8192), (1, 4096), (8, 16384), (32, 65536), (64, 3001), (100, 8192)])
def test_training_forward_parameter_gradients_match_autograd(cuda_device, in_dim, n, fused):
    """MLPEnergy.forward on an input that needs no gradient (the CD loss's data and negatives).
    fused = False: _ThinMLPEnergy -- the same forward ops (energies bit-identical to self.net), parameter gradients as row-block
    batched products instead of K = batch GEMMs.
    fused = True (the default where the kernels exist: hidden 64 / 128, in_dim <= 64): _FusedMLPTraining -- energies from the
    forward pass of the fused evaluation, the backward ONE launch (ebm_mlp_backward_acts_f32: seed-scaled backward on the matrix
    cores, the four activations stored hidden-major) + small-output products; kernel arithmetic, i.e. a tolerance tier.
    Bar: energies within 2e-5 (1 + |E|) of self.net; every parameter gradient no further from the fp64 gradient than autograd's own
    fp32 graph is (x 2 for the torch backward, x 4 for the kernel's)."""
    cpu, gpu = _models(cuda_device, in_dim, seed=7 + in_dim, scale=1.5)
    gpu.fused_training = fused
    x = torch.randn(n, in_dim, device=cuda_device)
    c0 = hip_calls("ebm_mlp_backward_acts_f32")
    e_fast = gpu(x)
    name = type(e_fast.grad_fn).__name__
    kernel_path = fused and in_dim <= 64
    assert name.startswith("_FusedMLPTraining" if kernel_path else "_ThinMLPEnergy"), name
    e_ref = gpu.net(x).squeeze(-1)
    if kernel_path:
        assert ((e_fast - e_ref).abs() / (1 + e_ref.abs())).max().item() <= 2e-5
    else:
        assert torch.equal(e_fast, e_ref)
    obj = lambda e: e.mean() + 0.1 * (e ** 2).mean()  # noqa: E731  (the shape of the CD loss)
    g_fast = torch.autograd.grad(obj(e_fast), list(gpu.parameters()))
    assert hip_calls("ebm_mlp_backward_acts_f32") == c0 + (1 if kernel_path else 0)
    g_auto = torch.autograd.grad(obj(e_ref), list(gpu.parameters()))
    m64 = copy.deepcopy(gpu).double()
    g_64 = torch.autograd.grad(obj(m64.net(x.double()).squeeze(-1)), list(m64.parameters()))
    for (pname, _), gf, ga, g6 in zip(gpu.named_parameters(), g_fast, g_auto, g_64):
        scale = g6.abs().max().item() + 1e-12
        err_fast = (gf.double() - g6).abs().max().item() / scale
        err_auto = (ga.double() - g6).abs().max().item() / scale
        assert gf.shape == ga.shape
        assert err_fast <= max((4.0 if kernel_path else 2.0) * err_auto, 4e-6 if kernel_path else 2e-6), (pname, err_fast, err_auto)
    # an input that needs its own gradient keeps autograd's graph through self.net (the samplers' step route, second derivatives)
    xr = x[:64].clone().requires_grad_(True)
    e = gpu(xr)
    assert not type(e.grad_fn).__name__.startswith(("_ThinMLPEnergy", "_FusedMLPTraining"))
    (gx,) = torch.autograd.grad(e.sum(), xr, create_graph=True)
    assert gx.requires_grad


def test_mlp_backward_acts_entry_against_torch(cuda_device):
    """ebm_mlp_backward_acts_f32 through the C ABI: the three stored planes (h1, the pre-activation a2, d1 of a unit seed), the
    energies and the seed-scaled input gradient against the same quantities from torch ops (hidden 64 and 128, a ragged row count,
    a non-trivial seed -- which scales grad_out only)."""
    for hidden, in_dim, n in ((128, 2, 1000), (64, 20, 333), (128, 64, 65), (64, 2, 200), (128, 1, 130)):
        torch.manual_seed(hidden + in_dim)
        model = ta.MLPEnergy(in_dim, hidden, device=cuda_device)
        spec = model.fused_spec()
        x = torch.randn(n, in_dim, device=cuda_device)
        seed = torch.randn(n, device=cuda_device)
        n_pad = (n + 127) // 128 * 128
        tiles = torch.full((n_pad // 32, 3, hidden, 32), float("nan"), device=cuda_device)  # tiles of 32 rows: [3][H][32] each
        e = torch.empty(n, device=cuda_device)
        g = torch.empty(n, in_dim, device=cuda_device)
        _lib.call("ebm_mlp_backward_acts_f32", spec.to_c(), x.data_ptr(), n, in_dim, seed.data_ptr(), e.data_ptr(), g.data_ptr(),
                  tiles.data_ptr(), _lib.stream_handle(cuda_device))
        acts = tiles.permute(1, 2, 0, 3).reshape(3, hidden, n_pad)
        net = model.net
        xr = x.clone().requires_grad_(True)
        a1 = net[0](xr); h1 = net[1](a1); a2 = net[2](h1); h2 = net[3](a2); en = net[4](h2).squeeze(-1)  # noqa: E702
        (d_a1,) = torch.autograd.grad(en, a1, grad_outputs=torch.ones_like(en), retain_graph=True)
        (d_x,) = torch.autograd.grad(en, xr, grad_outputs=seed)
        tol = dict(rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(e, en.detach(), rtol=2e-5, atol=2e-5)
        torch.testing.assert_close(acts[0, :, :n].t(), h1.detach(), **tol)
        torch.testing.assert_close(acts[1, :, :n].t(), a2.detach(), **tol)
        torch.testing.assert_close(acts[2, :, :n].t(), d_a1, **tol)
        torch.testing.assert_close(g, d_x, **tol)
        assert torch.isfinite(acts).all()  # the padding rows are written too (an all-zero input row's values; the gradient pass gives them seed 0)
    # shapes without the kernel
    wide = ta.MLPEnergy(100, 128, device=cuda_device).fused_spec()
    xx = torch.zeros(64, 100, device=cuda_device)
    aa = torch.empty(4, 3, 128, 32, device=cuda_device)
    with pytest.raises(RuntimeError, match="dim <= 64"):
        _lib.call("ebm_mlp_backward_acts_f32", wide.to_c(), xx.data_ptr(), 64, 100, None, None, None, aa.data_ptr(), _lib.stream_handle(cuda_device))


@pytest.mark.parametrize("hidden,in_dim,n", [(128, 2, 65536), (128, 2, 1000), (64, 20, 333), (128, 64, 4097), (64, 33, 8192), (128, 7, 32), (128, 32, 131072)])
@pytest.mark.parametrize("seeded", [True, False])
def test_mlp_param_grads_entry_against_fp64_products(cuda_device, hidden, in_dim, n, seeded):
    """ebm_mlp_param_grads_f32 through the C ABI (ABI 7): every parameter gradient from the stored planes h1 | a2 | d1 in one pass --
    h2 = silu(a2) and d2 = w3 silu'(a2) recomputed on load, fp32 MFMA products over K = n, partial records added in a fixed order.
    Referee: the same products in fp64 from the same planes; bar: the fp32 result within 4 x what torch's own fp32 ops make of it
    (and 4e-6 relative to the largest entry: the recomputation uses the hardware's exp2 / rcp), two launches bit-equal."""
    g = torch.Generator(device=cuda_device).manual_seed(hidden + in_dim + n)
    n_pad = (n + 127) // 128 * 128
    acts = torch.randn(3, hidden, n_pad, device=cuda_device, generator=g)
    acts[1] *= 2.0  # pre-activations over a few units
    tiles = acts.view(3, hidden, n_pad // 32, 32).permute(2, 0, 1, 3).contiguous()  # the layout of the entry: tiles of 32 rows
    x = torch.randn(n, in_dim, device=cuda_device, generator=g)
    w3 = torch.randn(hidden, device=cuda_device, generator=g)
    seed = torch.randn(n, device=cuda_device, generator=g) if seeded else None
    lib = _lib.lib()
    wf = int(lib.ebm_mlp_param_grads_work_f32(hidden, in_dim, n))
    assert wf > 0 and int(lib.ebm_mlp_param_grads_work_f32(256, in_dim, n)) == 0
    work = torch.empty(wf, device=cuda_device)
    sizes = (hidden * in_dim, hidden, hidden * hidden, hidden, hidden, 1)
    outs = []
    for _ in range(2):
        out = torch.full((sum(sizes),), float("nan"), device=cuda_device)
        work.normal_()  # the workspace needs no initialisation
        _lib.call("ebm_mlp_param_grads_f32", tiles.data_ptr(), n, hidden, x.data_ptr(), in_dim, seed.data_ptr() if seeded else None,
                  w3.data_ptr(), work.data_ptr(), wf, out.data_ptr(), _lib.stream_handle(cuda_device))
        outs.append(out)
    assert torch.equal(outs[0], outs[1])
    s32 = seed if seeded else torch.ones(n, device=cuda_device)
    def products(dt):
        h1, a2, d1 = (acts[i, :, :n].to(dt) for i in range(3))
        sg = torch.sigmoid(a2)
        h2 = a2 * sg
        d2 = w3.to(dt)[:, None] * (sg + h2 * (1 - sg))
        sd = s32.to(dt)
        return ((d1 * sd) @ x.to(dt), (d1 * sd).sum(1), (d2 * sd) @ h1.t(), (d2 * sd).sum(1), (h2 * sd).sum(1), sd.sum().reshape(1))
    got = outs[0].split(sizes)
    for name, gk, r32, r64 in zip(("dW1", "db1", "dW2", "db2", "dw3", "db3"), got, products(torch.float32), products(torch.float64)):
        scale = r64.abs().max().item() + 1e-30
        err = (gk.double() - r64.reshape(-1)).abs().max().item() / scale
        err32 = (r32.double() - r64).abs().max().item() / scale
        assert err <= max(4 * err32, 4e-6), (name, err, err32)
    with pytest.raises(ValueError, match="workspace"):
        _lib.call("ebm_mlp_param_grads_f32", tiles.data_ptr(), n, hidden, x.data_ptr(), in_dim, None, w3.data_ptr(), work.data_ptr(), wf - 1,
                  outs[0].data_ptr(), _lib.stream_handle(cuda_device))
    with pytest.raises(RuntimeError, match="dim <= 64"):
        _lib.call("ebm_mlp_param_grads_f32", tiles.data_ptr(), n, hidden, x.data_ptr(), 100, None, w3.data_ptr(), work.data_ptr(), wf,
                  outs[0].data_ptr(), _lib.stream_handle(cuda_device))
