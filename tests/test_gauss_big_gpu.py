"""Dense Gaussians at dims 132 .. 512 (multiples of 4): the tiled per-step kernel (csrc/gauss_big.hip) -- the state goes through
HBM once per step, Ps through LDS as three bf16 splits per K-slab, the update is the contraction's epilogue.  Checked against
the oracle's chain (reference op order, fp32) and an fp64 referee on the same injected noise, and -- native RNG -- against
the kernel fed the materialised Philox field (bit for bit: same arithmetic, the field depends on (seed, step, element) only)."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients

pytestmark = pytest.mark.gpu


def _model(dim, device, seed=0):
    g = torch.Generator().manual_seed(seed + dim)
    a = torch.randn(dim, dim, generator=g)
    cov = a @ a.t() / dim + 0.5 * torch.eye(dim)
    mean = torch.randn(dim, generator=g) * 0.5
    return ta.GaussianModel(mean, cov, device=device), oracle.Gaussian(mean, cov)


def _call(spec, x, k, rows, clamp=None, thin=1, traj=None, noise=None, seed=0, step=0):
    n, dim = x.shape
    a, sq, coef = rows[0]
    table = None
    if len(rows) > 1:
        table = torch.tensor([(r[0], r[1], r[2], 0.0) for r in rows], dtype=torch.float32, device=x.device)
    clamp_on, cmin, cmax = (0, 0.0, 0.0) if clamp is None else (1, clamp[0], clamp[1])
    _lib.call("ebm_langevin_chain_f32", spec.to_c(), x.data_ptr(), n, dim, k, a, sq, coef, _lib.ptr(table), clamp_on, cmin, cmax,
              thin, _lib.ptr(traj), None, _lib.ptr(noise), seed, step, _lib.stream_handle(x.device))


def _f64_chain(model, x0, noise, etas, sigmas):
    prec = model.cov_inv.double()
    mean = model.mean.double()
    x = x0.double()
    for i, (eta, sigma) in enumerate(zip(etas, sigmas)):
        x = x - eta * ((x - mean) @ prec) + sigma * (noise[i].double() * eta ** 0.5)
    return x


@pytest.mark.parametrize("dim,n", [(132, 300), (160, 257), (200, 128), (256, 515), (260, 200), (320, 129), (384, 130), (500, 77), (512, 260)])
def test_injected_noise_matches_the_oracle_and_the_fp64_referee(cuda_device, dim, n):
    model, ref = _model(dim, cuda_device)
    k = 6
    etas = [0.02 * (0.9 ** i) for i in range(k)]
    sigmas = [1.0 - 0.05 * i for i in range(k)]
    gen = torch.Generator().manual_seed(dim)
    x0 = torch.randn(n, dim, generator=gen)
    noise = torch.randn(k, n, dim, generator=gen)
    want, wtraj, _ = oracle.langevin_chain(ref, x0, noise, etas, sigmas, thin=2, want_traj=True)
    f64 = _f64_chain(ref, x0, noise, etas, sigmas)
    x = x0.to(cuda_device)
    traj = torch.full((n, k // 2, dim), float("nan"), device=cuda_device)
    rows = [em_coefficients(e, s) for e, s in zip(etas, sigmas)]
    _call(model.fused_spec(), x, k, rows, thin=2, traj=traj, noise=noise.to(cuda_device))
    got = x.cpu()
    # the oracle itself (torch fp32 bmm) is a few ulp of |P||d| from the fp64 chain; the split-operand contraction is the same class
    err_ref = (want.double() - f64).abs().max().item()
    err_hip = (got.double() - f64).abs().max().item()
    assert err_hip <= 4.0 * err_ref + 1e-6, (err_hip, err_ref)
    torch.testing.assert_close(got, want, rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(traj.cpu(), wtraj, rtol=2e-5, atol=2e-5)
    assert torch.equal(traj[:, -1].cpu(), got)


@pytest.mark.parametrize("dim,n", [(160, 1000), (256, 300), (448, 257)])
def test_native_rng_equals_the_materialised_field_and_clamp(cuda_device, dim, n):
    model, _ = _model(dim, cuda_device, seed=3)
    k, seed, step0 = 5, 0x1234ABCD77, 40
    rows = [em_coefficients(0.01, 1.0)]
    x0 = torch.randn(n, dim, device=cuda_device) * 2.0
    noise = torch.empty(k, n, dim, device=cuda_device)
    for i in range(k):
        _lib.call("ebm_noise_fill_f32", noise[i].data_ptr(), n * dim, _lib.NOISE_NORMAL, seed, step0 + i, _lib.stream_handle(cuda_device))
    for clamp in (None, (-1.5, 1.5)):
        xa, xb = x0.clone(), x0.clone()
        _call(model.fused_spec(), xa, k, rows, clamp=clamp, seed=seed, step=step0)
        _call(model.fused_spec(), xb, k, rows, clamp=clamp, noise=noise)
        assert torch.equal(xa, xb)
        assert not torch.equal(xa, x0)
        if clamp:
            assert xa.min().item() >= -1.5 and xa.max().item() <= 1.5


def test_sampler_route_is_one_launch_and_recovers_the_moments(cuda_device):
    dim, n = 256, 8192
    g = torch.Generator().manual_seed(1)
    q, _ = torch.linalg.qr(torch.randn(dim, dim, generator=g))
    var = torch.linspace(0.5, 2.0, dim)
    cov = (q * var) @ q.t()
    mean = torch.randn(dim, generator=g)
    model = ta.GaussianModel(mean, cov, device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.05, device=cuda_device)
    c0 = hip_calls("ebm_langevin_chain_f32")
    out = s.sample(x=torch.randn(n, dim, device=cuda_device), n_steps=400, generator=torch.Generator(device=cuda_device).manual_seed(2))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
    # Euler-Maruyama's stationary law is N(mean, cov (I - eta P / 2)^-1): the discretisation inflates cov by <= 5 % here
    emp_mean = out.mean(dim=0).cpu()
    emp_cov = torch.cov(out.t().cpu())
    assert (emp_mean - mean).abs().max().item() < 0.08
    rel = torch.linalg.norm(emp_cov - cov) / torch.linalg.norm(cov)
    assert rel.item() < 0.2, rel.item()


def test_dims_it_does_not_take_keep_the_lane_group_kernel(cuda_device):
    for dim in (130, 516):
        model, ref = _model(dim, cuda_device)
        x0 = torch.randn(64, dim)
        noise = torch.randn(2, 64, dim)
        want, _, _ = oracle.langevin_chain(ref, x0, noise, [0.01, 0.01], [1.0, 1.0])
        x = x0.to(cuda_device)
        _call(model.fused_spec(), x, 2, [em_coefficients(0.01, 1.0)], noise=noise.to(cuda_device))
        torch.testing.assert_close(x.cpu(), want, rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("dim,n", [(160, 1000), (256, 515), (384, 130)])
def test_diagnostics_come_from_in_kernel_records(cuda_device, dim, n):
    """The streamed-Ps kernels keep records like the other matrix-layout kernels (one per wave-tile of 32 chains; the energy
    share one step late, a kept last step one contraction more): return_diagnostics=True is ONE chain launch, the chains are
    the same as without, and the numbers are those of torch reductions over the stored trajectory."""
    model, _ = _model(dim, cuda_device, seed=5)
    layout = _lib.diag_layout(model.fused_spec().to_c(), _lib.DIAG_LANGEVIN, n, dim)
    assert layout == ((n + 31) // 32, dim, 32 * dim)
    s = ta.LangevinDynamics(model, step_size=0.02, device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device)
    for n_steps, thin in ((12, 4), (13, 4), (6, 1)):
        c0 = hip_calls("ebm_langevin_chain_f32")
        traj, diag = s.sample(x=x0, n_steps=n_steps, thin=thin, return_trajectory=True, return_diagnostics=True,
                              generator=torch.Generator(device=cuda_device).manual_seed(3))
        assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
        plain = s.sample(x=x0, n_steps=n_steps, generator=torch.Generator(device=cuda_device).manual_seed(3))
        kept_only = s.sample(x=x0, n_steps=(n_steps // thin) * thin, generator=torch.Generator(device=cuda_device).manual_seed(3))
        assert torch.equal(traj[:, -1], kept_only)
        _, diag2 = s.sample(x=x0, n_steps=n_steps, thin=thin, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(3))
        assert plain.shape == x0.shape and torch.equal(diag2["mean"], diag["mean"])
        t64 = traj.double()
        torch.testing.assert_close(diag["mean"].double(), t64.mean(dim=0), rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(diag["var"].double(), t64.var(dim=0, unbiased=False), rtol=1e-4, atol=1e-7)
        want_e = torch.stack([model(traj[:, j]).double().mean() for j in range(n_steps // thin)])
        torch.testing.assert_close(diag["energy"].double(), want_e, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("dim", [258, 1024])
def test_widths_without_a_matrix_chain_kernel_take_the_gemm_step_route(cuda_device, dim):
    """dim 258 (not a multiple of 4 and beyond the shifted-row kernels, tests/test_gauss_shift_gpu.py) / 1024 (above 512): the sampler runs the per-step route -- the gradient as one library
    GEMM (GaussianModel's closed form above 128 dims), the update kernel on the native field -- not the lane-group chain
    kernel; same chains as that kernel on the same seed (shared (seed, step, element) field), to fp32 round-off.  The reroute
    is for batches that fill the GEMM and calls that can be replayed from a graph: few chains, or
    capture_graph = False, keep the ONE fused launch."""
    n, k = 16384, 5
    model, _ = _model(dim, cuda_device, seed=7)
    s = ta.LangevinDynamics(model, step_size=0.01, device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device)
    c_chain = hip_calls("ebm_langevin_chain_f32")
    s.sample(x=x0[:512], n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(11))
    s.capture_graph = False
    s.sample(x=x0, n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(11))
    s.capture_graph = None
    assert hip_calls("ebm_langevin_chain_f32") == c_chain + 2
    c_chain, c_step = hip_calls("ebm_langevin_chain_f32"), hip_calls("ebm_langevin_step_f32") + hip_calls("ebm_langevin_step_dev_f32")
    out = s.sample(x=x0, n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(11))
    assert hip_calls("ebm_langevin_chain_f32") == c_chain
    assert hip_calls("ebm_langevin_step_f32") + hip_calls("ebm_langevin_step_dev_f32") > c_step
    # the closed-form gradient against autograd through the reference's forward
    g_fast = model.gradient(x0)
    leaf = x0.clone().requires_grad_(True)
    (g_auto,) = torch.autograd.grad(model(leaf).sum(), leaf)
    torch.testing.assert_close(g_fast, g_auto, rtol=2e-5, atol=2e-5)
    # the lane-group chain kernel on the same seed / offsets
    from torchebm_amd import _rng
    seed, first = _rng.reserve(torch.Generator(device=cuda_device).manual_seed(11), cuda_device, k)
    xb = x0.clone()
    _call(model.fused_spec(), xb, k, [em_coefficients(0.01, 1.0)], seed=seed, step=first)
    torch.testing.assert_close(out, xb, rtol=5e-5, atol=5e-5)


def test_wide_gaussian_hmc_takes_the_gemm_transition_route(cuda_device):
    """Above 256 dims (round 4: 164 .. 256 run the streamed transition kernel, tests/test_gauss_hmc_stream_gpu.py)
    HamiltonianMonteCarlo runs the per-transition route for a GaussianModel -- gradient and energy as one
    library GEMM each, kick / drift / accept kernels on the native field -- instead of the lane-group transition kernel;
    same generator => same draws, so the chains agree with that kernel's (through the C ABI) except for borderline accepts."""
    dim, n, T, L = 320, 16384, 4, 5
    model, _ = _model(dim, cuda_device, seed=9)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.08, n_leapfrog_steps=L, device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device)
    c0 = hip_calls("ebm_hmc_chain_f32")
    s.sample(x=x0[:512], n_steps=T, generator=torch.Generator(device=cuda_device).manual_seed(21))  # few chains: the fused launch
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    c0 += 1
    out, diag = s.sample(x=x0, n_steps=T, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(21))
    assert hip_calls("ebm_hmc_chain_f32") == c0
    assert 0.5 < diag["acceptance_rate"].mean().item() <= 1.0
    # the forward the route uses (one GEMM) against the reference's batched form
    x = out
    delta = x - model.mean
    batched = 0.5 * torch.bmm(delta.unsqueeze(1), torch.bmm(model.cov_inv.unsqueeze(0).expand(n, -1, -1), delta.unsqueeze(-1))).squeeze()
    torch.testing.assert_close(model(x), batched, rtol=2e-5, atol=2e-5)
    # the transition kernel on the same seed
    from torchebm_amd import _rng
    seed, first = _rng.reserve(torch.Generator(device=cuda_device).manual_seed(21), cuda_device, 2 * T)
    xb = x0.clone()
    mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
    _lib.call("ebm_hmc_chain_f32", model.fused_spec().to_c(), xb.data_ptr(), n, dim, T, L, 0.08, None, 0, 1.0, None, 1, None, None,
              mask.data_ptr(), None, None, None, seed, first, _lib.stream_handle(cuda_device))
    rows_equal = ((out - xb).abs().max(dim=1).values < 2e-3).float().mean().item()
    assert rows_equal > 0.97, rows_equal


@pytest.mark.parametrize("seed", range(6))
def test_random_shapes_schedules_thinning_and_clamp_against_the_oracle(cuda_device, seed):
    """Random widths (multiples of 4 in 132 .. 512: all three kernel families, ragged last tiles), chain counts with ragged
    last workgroups, scheduled coefficients, thinning with a trajectory, and a clamp that bites -- the oracle's chain on the
    same injected noise."""
    g = torch.Generator().manual_seed(1000 + seed)
    dim = 4 * int(torch.randint(33, 129, (1,), generator=g))
    n = int(torch.randint(1, 400, (1,), generator=g))
    k = int(torch.randint(3, 9, (1,), generator=g))
    thin = int(torch.randint(1, 4, (1,), generator=g))
    model, ref = _model(dim, cuda_device, seed=seed)
    etas = [0.02 * (0.85 ** i) for i in range(k)]
    sigmas = [1.0 - 0.07 * i for i in range(k)]
    clamp = (-1.2, 1.4) if seed % 2 else None
    x0 = torch.randn(n, dim, generator=g) * 1.5
    noise = torch.randn(k, n, dim, generator=g)
    want, wtraj, _ = oracle.langevin_chain(ref, x0, noise, etas, sigmas, clamp=clamp, thin=thin, want_traj=True)
    x = x0.to(cuda_device)
    n_kept = k // thin
    traj = torch.full((n, n_kept, dim), float("nan"), device=cuda_device) if n_kept else None
    rows = [em_coefficients(e, s) for e, s in zip(etas, sigmas)]
    _call(model.fused_spec(), x, k, rows, clamp=clamp, thin=thin, traj=traj, noise=noise.to(cuda_device))
    torch.testing.assert_close(x.cpu(), want, rtol=3e-5, atol=3e-5)
    if n_kept:
        torch.testing.assert_close(traj.cpu(), wtraj, rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("dim,n", [(132, 200), (160, 257), (256, 515), (320, 129), (500, 77), (512, 260)])
def test_energy_and_gradient_entry_on_the_matrix_cores(cuda_device, dim, n):
    """ebm_energy_grad_f32 above 128 dims: one contraction pass of the tiled kernel (nothing updated) -- against fp64 with the
    fp32 torch forms as the yardstick."""
    model, ref = _model(dim, cuda_device, seed=11)
    x = torch.randn(n, dim, device=cuda_device) * 1.3
    x_before = x.clone()
    e = torch.full((n,), float("nan"), device=cuda_device)
    g = torch.full((n, dim), float("nan"), device=cuda_device)
    st = _lib.stream_handle(cuda_device)
    _lib.call("ebm_energy_grad_f32", model.fused_spec().to_c(), x.data_ptr(), n, dim, e.data_ptr(), g.data_ptr(), st)
    assert torch.equal(x, x_before)
    d64 = x.double() - model.mean.double()
    p64 = 0.5 * (model.cov_inv.double() + model.cov_inv.double().t())
    g64 = d64 @ p64
    e64 = 0.5 * (d64 * g64).sum(dim=1)
    g32 = (x - model.mean) @ (0.5 * (model.cov_inv + model.cov_inv.t()))
    err_ref = (g32.double() - g64).abs().max().item()
    assert (g.double() - g64).abs().max().item() <= 4.0 * err_ref + 1e-6
    torch.testing.assert_close(e.double(), e64, rtol=2e-5, atol=2e-4)
    # either output alone
    e2 = torch.empty_like(e)
    _lib.call("ebm_energy_grad_f32", model.fused_spec().to_c(), x.data_ptr(), n, dim, e2.data_ptr(), None, st)
    g2 = torch.empty_like(g)
    _lib.call("ebm_energy_grad_f32", model.fused_spec().to_c(), x.data_ptr(), n, dim, None, g2.data_ptr(), st)
    assert torch.equal(e2, e) and torch.equal(g2, g)


@pytest.mark.parametrize("dim,mass", [(132, None), (160, None), (160, 1.7), (148, "diag")])
def test_hmc_five_tile_transition_kernel_against_the_oracle(cuda_device, dim, mass):
    """HMC at dims 132 .. 160: five tiles -- the three split images of Ps (150 KB) still fit the CU's LDS -- on the matrix-core
    transition kernel; injected momenta / uniforms, accept decisions identical to the oracle's wherever its margin is not
    borderline, states to the Gaussian tolerance; with and without records."""
    from torchebm_amd.integrators.symplectic import _mass_args
    n, T, L, eps = 300, 4, 5, 0.06
    model, ref = _model(dim, cuda_device, seed=13)
    g = torch.Generator().manual_seed(dim)
    x0 = torch.randn(n, dim, generator=g)
    p = torch.randn(T, n, dim, generator=g)
    u = torch.rand(T, n, generator=g)
    m = None
    if mass == "diag":
        m = torch.rand(dim, generator=g) + 0.5
    elif mass is not None:
        m = mass
    o = oracle.hmc_chain(ref, x0, p, u, [eps] * T, L, mass=m, thin=2, want_diag=True, want_margins=True)
    m_dev = m.to(cuda_device) if torch.is_tensor(m) else m
    p_dev, u_dev = p.to(cuda_device), u.to(cuda_device)  # (kept referenced until the launches that read them are enqueued)
    for with_records in (False, True):
        x = x0.to(cuda_device)
        mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
        kind, ms, md = _mass_args(m_dev, x)
        desc = model.fused_spec().to_c()
        rec = None
        if with_records:
            nb, S, E = _lib.diag_layout(desc, _lib.DIAG_HMC, n, dim, True, False)
            rec = torch.empty((T // 2) * nb * (2 * S + 8), device=cuda_device)
        _lib.call("ebm_hmc_chain_f32", desc, x.data_ptr(), n, dim, T, L, eps, None, kind, ms, _lib.ptr(md), 2, None, _lib.ptr(rec),
                  mask.data_ptr(), None, p_dev.data_ptr(), u_dev.data_ptr(), 0, 0, _lib.stream_handle(cuda_device))
        safe = o["margins"].abs() > 1e-4
        assert torch.equal(mask.cpu().bool()[safe], o["accepted"][safe])
        rows_ok = (mask.cpu().bool() == o["accepted"]).all(dim=0)
        torch.testing.assert_close(x.cpu()[rows_ok], o["x"][rows_ok], rtol=5e-4, atol=5e-4)
    # and the sampler takes it (one launch)
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, mass=m_dev, device=cuda_device)
    c0 = hip_calls("ebm_hmc_chain_f32")
    s.sample(x=x0.to(cuda_device), n_steps=T)
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
