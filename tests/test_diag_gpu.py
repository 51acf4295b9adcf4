"""In-kernel sampler diagnostics (include/ebm_hip.h: ``diag_partials``, ``ebm_diag_layout``,
``ebm_diag_finish_f32``; reference: samplers/langevin_dynamics.py:170-185, samplers/hmc.py:294-310).

``return_diagnostics=True`` on the fused routes is ONE chain launch + one merge launch: every workgroup stores a
record of its chains' partial sums at each kept step.  Checked here against (a) the oracle's diagnostics on the
materialised noise field and (b) fp64 torch reductions of the trajectory the same call returned, over every record
geometry: flat 1024-element blocks (dim | 1024, 1024 | dim), lane-group blocks of whole rows (ragged dims, coupled
energies, HMC), partial last blocks, a single chain, populations far from the origin, and chunked record buffers."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib, _rng

pytestmark = pytest.mark.gpu


def _check_against_trajectory(model, traj, diag, rtol=2e-5, e_rtol=2e-5, accept=None):
    n, n_kept = traj.shape[0], traj.shape[1]
    for j in range(n_kept):
        xs = traj[:, j].double()
        torch.testing.assert_close(diag["mean"][j].double(), xs.mean(dim=0), rtol=rtol, atol=2e-6)
        want_var = xs.var(dim=0, unbiased=False).clamp(1e-10, 1e10) if n > 1 else torch.zeros_like(xs[0])
        torch.testing.assert_close(diag["var"][j].double(), want_var, rtol=5 * rtol, atol=1e-7)
        e = model(traj[:, j]).double()
        if accept is not None:
            e = e.clamp(-1e10, 1e10)
        torch.testing.assert_close(diag["energy"][j].double(), e.mean(), rtol=e_rtol, atol=1e-4)


@pytest.mark.parametrize("dim,n", [(1, 3001), (2, 1500), (4, 777), (8, 4099), (64, 1000), (128, 300), (256, 37), (512, 21),
                                   (1024, 9), (2048, 5), (4096, 3), (3, 1000), (7, 333), (100, 1000), (700, 40)])
def test_langevin_elementwise_records_every_geometry(cuda_device, dim, n):
    model = ta.DoubleWellModel(device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.005, device=cuda_device)
    x0 = torch.randn(n, dim, generator=torch.Generator().manual_seed(dim)).clamp_(-2.0, 2.0)
    k, thin = 14, 3
    c0, f0, st0 = hip_calls("ebm_langevin_chain_f32"), hip_calls("ebm_diag_finish_f32"), hip_calls("ebm_chain_stats_f32")
    traj, diag = s.sample(x=x0.to(cuda_device), n_steps=k, thin=thin, return_trajectory=(dim % 4 == 0 or 1024 % dim != 0),
                          return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(11))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 1 and hip_calls("ebm_diag_finish_f32") == f0 + 1  # ONE chain launch
    assert hip_calls("ebm_chain_stats_f32") == st0  # no pass over the state
    # the oracle on the field the kernel drew
    rows = []
    for i in range(k):  # one allocation per step: the ABI wants 16-byte aligned pointers
        buf = torch.empty(n, dim, device=cuda_device)
        _lib.call("ebm_noise_fill_f32", buf.data_ptr(), n * dim, _lib.NOISE_NORMAL, _rng.kernel_seed(11), i,
                  _lib.stream_handle(cuda_device))
        rows.append(buf)
    noise = torch.stack(rows)
    wx, wtraj, wdiag = oracle.langevin_chain(oracle.DoubleWell(), x0, noise.cpu(), [0.005] * k, [1.0] * k, thin=thin,
                                             want_traj=True, want_diag=True)
    torch.testing.assert_close(diag["mean"].cpu(), wdiag["mean"], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(diag["var"].cpu(), wdiag["var"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(diag["energy"].cpu(), wdiag["energy"], rtol=1e-5, atol=1e-4)
    if traj.ndim == 3:
        assert torch.equal(traj.cpu(), wtraj)  # element-wise chains stay bit-exact with records switched on
    else:
        assert torch.equal(traj.cpu(), wx)


def test_records_do_not_change_the_chains_and_chunking_is_invisible(cuda_device):
    model = ta.DoubleWellModel(device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=ta.core.LinearScheduler(0.01, 0.002, 10), clamp=(-2.5, 2.5), device=cuda_device)
    x0 = torch.randn(5000, 64, device=cuda_device).clamp_(-2.0, 2.0)
    gen = lambda: torch.Generator(device=cuda_device).manual_seed(3)  # noqa: E731
    plain = s.sample(x=x0, n_steps=23, thin=4, return_trajectory=True, generator=gen())
    traj, diag = s.sample(x=x0, n_steps=23, thin=4, return_trajectory=True, return_diagnostics=True, generator=gen())
    assert torch.equal(plain, traj)
    _check_against_trajectory(model, traj, diag)
    # force the record buffer to hold one kept step at a time: 5 chain launches, same bits
    s.DIAG_RECORD_BYTES = 1
    c0 = hip_calls("ebm_langevin_chain_f32")
    traj2, diag2 = s.sample(x=x0, n_steps=23, thin=4, return_trajectory=True, return_diagnostics=True, generator=gen())
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 5
    assert torch.equal(traj2, traj) and all(torch.equal(diag2[k_], diag[k_]) for k_ in diag)
    # no trajectory, final state only
    fin, diag3 = s.sample(x=x0, n_steps=23, thin=4, return_diagnostics=True, generator=gen())
    assert torch.equal(fin, s.sample(x=x0, n_steps=23, generator=gen())) and torch.equal(diag3["mean"], diag["mean"])


def test_population_far_from_the_origin(cuda_device):
    """mean ~ 1000, std ~ 0.05: the merge works on block-local second moments, so the variance keeps its digits
    (a plain sum-of-squares formula in fp32 would return noise here)."""
    model = ta.HarmonicModel(k=1.0, device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=1e-6, noise_scale=1.0, device=cuda_device)
    x0 = (1000.0 + 0.05 * torch.randn(20000, 16, generator=torch.Generator().manual_seed(0))).to(cuda_device)
    traj, diag = s.sample(x=x0, n_steps=4, thin=2, return_trajectory=True, return_diagnostics=True)
    for j in range(2):
        xs = traj[:, j].double()
        torch.testing.assert_close(diag["mean"][j].double(), xs.mean(dim=0), rtol=2e-7, atol=0)
        torch.testing.assert_close(diag["var"][j].double(), xs.var(dim=0, unbiased=False), rtol=2e-3, atol=0)


def test_single_chain_and_tiny_populations(cuda_device):
    model = ta.DoubleWellModel(device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.01, device=cuda_device)
    x0 = torch.randn(1, 8, device=cuda_device)
    traj, diag = s.sample(x=x0, n_steps=6, thin=2, return_trajectory=True, return_diagnostics=True)
    assert torch.equal(diag["mean"], traj[0]) and torch.count_nonzero(diag["var"]) == 0  # langevin_dynamics.py:179-181
    torch.testing.assert_close(diag["energy"], torch.stack([model(traj[:, j]).mean() for j in range(3)]), rtol=1e-5, atol=1e-5)
    h = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=3, device=cuda_device)
    traj, diag = h.sample(x=x0, n_steps=4, return_trajectory=True, return_diagnostics=True)
    assert torch.equal(diag["mean"], traj[0]) and torch.count_nonzero(diag["var"]) == 0  # hmc.py:300-303
    for n in (2, 3, 65):
        traj, diag = s.sample(x=torch.randn(n, 4, device=cuda_device), n_steps=3, return_trajectory=True, return_diagnostics=True)
        _check_against_trajectory(model, traj, diag)


@pytest.mark.parametrize("kind,dim", [("gauss", 2), ("gauss", 8), ("gauss", 64), ("gauss", 100), ("gmm", 6), ("gmm", 16), ("gmm", 32),
                                      ("gmm40", 12)])
@pytest.mark.parametrize("heun", [False, True])
def test_langevin_coupled_energies(cuda_device, kind, dim, heun):
    g = torch.Generator().manual_seed(dim)
    if kind == "gauss":
        a = torch.randn(dim, dim, generator=g)
        model = ta.GaussianModel(torch.randn(dim, generator=g), a @ a.t() / dim + 0.5 * torch.eye(dim), device=cuda_device)
    else:
        k_comp = 40 if kind == "gmm40" else 5
        model = ta.GaussianMixtureModel(torch.randn(k_comp, dim, generator=g) * 2, sigma=0.9, device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.02, noise_scale=0.8, integrator="heun" if heun else None, device=cuda_device)
    n = 777
    x0 = torch.randn(n, dim, generator=g).to(cuda_device)
    entry = "ebm_langevin_heun_chain_f32" if heun else "ebm_langevin_chain_f32"
    c0 = hip_calls(entry)
    traj, diag = s.sample(x=x0, n_steps=9, thin=2, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(1))
    assert hip_calls(entry) == c0 + 1
    _check_against_trajectory(model, traj, diag, e_rtol=1e-4)
    # the chains are those of the call without diagnostics, to the tolerance of the two Gaussian kernels
    # (with records the lane-group kernel runs where the MFMA kernel would: another summation order)
    plain = s.sample(x=x0, n_steps=9, thin=2, return_trajectory=True, generator=torch.Generator(device=cuda_device).manual_seed(1))
    torch.testing.assert_close(plain, traj, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kind,dim,mass", [("dw", 2, None), ("dw", 32, None), ("dw", 64, 2.0), ("dw", 100, "diag"), ("dw", 1000, None),
                                           ("har", 7, None), ("gauss", 8, None), ("gauss", 64, None), ("gmm8", 32, None),
                                           ("gmm8", 32, "diag"), ("gmm5", 6, 2.0), ("gmm8", 16, None)])
def test_hmc_records(cuda_device, kind, dim, mass):
    g = torch.Generator().manual_seed(dim + 1)
    if kind == "dw":
        model, en = ta.DoubleWellModel(device=cuda_device), oracle.DoubleWell()
    elif kind == "har":
        model, en = ta.HarmonicModel(1.3, device=cuda_device), oracle.Harmonic(1.3)
    elif kind == "gauss":
        a = torch.randn(dim, dim, generator=g)
        mean, cov = torch.randn(dim, generator=g), a @ a.t() / dim + 0.5 * torch.eye(dim)
        model, en = ta.GaussianModel(mean, cov, device=cuda_device), oracle.Gaussian(mean, cov)
    else:
        means = torch.randn(int(kind[3:]), dim, generator=g) * 1.5
        model, en = ta.GaussianMixtureModel(means, sigma=1.1, device=cuda_device), oracle.GaussianMixture(means, 1.1)
    if mass == "diag":
        mass = torch.rand(dim, generator=g) + 0.5
    n, T, L, eps, thin = (90 if dim >= 1000 else 600), 7, 4, 0.03, 2
    x0 = torch.randn(n, dim, generator=g).clamp_(-1.5, 1.5)
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, mass=mass.to(cuda_device) if torch.is_tensor(mass) else mass,
                                 device=cuda_device)
    c0, f0 = hip_calls("ebm_hmc_chain_f32"), hip_calls("ebm_diag_finish_f32")
    traj, diag = s.sample(x=x0.to(cuda_device), n_steps=T, thin=thin, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(5))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1 and hip_calls("ebm_diag_finish_f32") == f0 + 1
    assert set(diag) == {"mean", "var", "energy", "acceptance_rate"} and diag["acceptance_rate"].shape == (T // thin,)
    _check_against_trajectory(model, traj, diag, e_rtol=1e-4, accept=True)
    # the oracle on the materialised draws: acceptance rate of the kept transitions (exact unless a call is borderline)
    st = _lib.stream_handle(cuda_device)
    p, u = torch.empty(T, n, dim, device=cuda_device), torch.empty(T, n, device=cuda_device)
    for t in range(T):
        _lib.call("ebm_noise_fill_f32", p[t].data_ptr(), n * dim, _lib.NOISE_NORMAL, _rng.kernel_seed(5), 2 * t, st)
        ut = torch.empty(n, device=cuda_device)
        _lib.call("ebm_noise_fill_f32", ut.data_ptr(), n, _lib.NOISE_UNIFORM, _rng.kernel_seed(5), 2 * t + 1, st)
        u[t] = ut
    want = oracle.hmc_chain(en, x0, p.cpu(), u.cpu(), [eps] * T, L, mass=mass, thin=thin, want_diag=True)
    if want["margin"] > 1e-4:
        torch.testing.assert_close(diag["acceptance_rate"].cpu(), want["diagnostics"]["acceptance_rate"])
        torch.testing.assert_close(diag["mean"].cpu(), want["diagnostics"]["mean"], rtol=1e-3, atol=1e-4)
        torch.testing.assert_close(diag["energy"].cpu(), want["diagnostics"]["energy"], rtol=1e-3, atol=1e-3)
    # without diagnostics the same chains
    plain = s.sample(x=x0.to(cuda_device), n_steps=T, thin=thin, return_trajectory=True,
                     generator=torch.Generator(device=cuda_device).manual_seed(5))
    # (with records the lane-group kernel runs; without, dense Gaussians and mixtures at dims 20 .. 96 take the
    #  matrix-layout kernels: the same chains to the tolerance tier, not bit for bit)
    matrix_route = kind == "gauss" or (kind.startswith("gmm") and 20 <= dim <= 96 and dim % 4 == 0
                                       and not (dim == 32 and int(kind[3:]) <= 8))
    if not matrix_route:
        assert torch.equal(plain, traj)
    else:
        err = ((plain - traj).abs() / traj.abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
        assert (err <= 5e-4).float().mean().item() >= 0.97, err.max().item()


@pytest.mark.parametrize("kind,dim", [("gauss", 20), ("gauss", 32), ("gauss", 64), ("gauss", 96), ("gmm16", 32), ("gmm16", 64),
                                      ("gmm5", 48), ("gmm32", 96)])
def test_matrix_layout_kernels_emit_records(cuda_device, kind, dim):
    """Dense Gaussians and mixtures where the matrix-layout Langevin kernels run (dims 20 .. 96): the records come from
    those kernels -- the layout query says 128 chains per workgroup, the chains are bit for bit those of the call without
    diagnostics, statistics and energy agree with the kept states."""
    g = torch.Generator().manual_seed(dim)
    if kind == "gauss":
        a = torch.randn(dim, dim, generator=g)
        model = ta.GaussianModel(torch.randn(dim, generator=g), a @ a.t() / dim + 0.5 * torch.eye(dim), device=cuda_device)
    else:
        model = ta.GaussianMixtureModel(torch.randn(int(kind[3:]), dim, generator=g) * 2, sigma=0.9, device=cuda_device)
    n = 1001
    assert _lib.diag_layout(model.fused_spec().to_c(), _lib.DIAG_LANGEVIN, n, dim) == ((n + 31) // 32, dim, 32 * dim)  # one record per wave of 32 chains
    s = ta.LangevinDynamics(model, step_size=0.02, noise_scale=0.8, clamp=(-3.0, 3.5), device=cuda_device)
    x0 = torch.randn(n, dim, generator=g).to(cuda_device)
    c0 = hip_calls("ebm_langevin_chain_f32")
    traj, diag = s.sample(x=x0, n_steps=9, thin=2, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(1))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
    _check_against_trajectory(model, traj, diag, e_rtol=1e-4)
    plain = s.sample(x=x0, n_steps=9, thin=2, return_trajectory=True, generator=torch.Generator(device=cuda_device).manual_seed(1))
    assert torch.equal(plain, traj)
    # and without a trajectory, a single chain per workgroup tail
    out, diag2 = s.sample(x=x0[:130], n_steps=4, thin=4, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(2))
    torch.testing.assert_close(diag2["mean"][0].double(), out.double().mean(0), rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(diag2["energy"][0].double(), model(out).double().mean(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("kind,dim,mass", [("gauss", 32, None), ("gauss", 64, 1.5), ("gauss", 96, "diag"), ("gauss", 96, None),
                                           ("gmm16", 32, None), ("gmm9", 64, "diag"), ("gmm32", 48, 0.7), ("gmm12", 96, "diag")])
def test_matrix_layout_hmc_kernels_emit_records(cuda_device, kind, dim, mass):
    """HMC with diagnostics where the matrix-layout kernels run and their layout does not depend on the mass form (dense
    Gaussians and mixtures at dims 20 .. 96): records from those kernels -- 128 chains per workgroup, the
    chains bit for bit those of the call without diagnostics, statistics / energy / acceptance rate of the kept states."""
    g = torch.Generator().manual_seed(dim)
    if kind == "gauss":
        a = torch.randn(dim, dim, generator=g)
        model = ta.GaussianModel(torch.randn(dim, generator=g), a @ a.t() / dim + 0.5 * torch.eye(dim), device=cuda_device)
    else:
        model = ta.GaussianMixtureModel(torch.randn(int(kind[3:]), dim, generator=g) * 1.5, sigma=1.1, device=cuda_device)
    if mass == "diag":
        mass = torch.rand(dim, generator=g) + 0.5
    n, T, L, thin = 700, 6, 4, 2
    assert _lib.diag_layout(model.fused_spec().to_c(), _lib.DIAG_HMC, n, dim) == ((n + 31) // 32, dim, 32 * dim)  # one record per wave of 32 chains
    s = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=L, mass=mass.to(cuda_device) if torch.is_tensor(mass) else mass,
                                 device=cuda_device)
    x0 = torch.randn(n, dim, generator=g).clamp_(-1.5, 1.5).to(cuda_device)
    c0 = hip_calls("ebm_hmc_chain_f32")
    traj, diag = s.sample(x=x0, n_steps=T, thin=thin, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(5))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    _check_against_trajectory(model, traj, diag, e_rtol=1e-4, accept=True)
    plain = s.sample(x=x0, n_steps=T, thin=thin, return_trajectory=True, generator=torch.Generator(device=cuda_device).manual_seed(5))
    assert torch.equal(plain, traj)
    # acceptance rate of a kept transition = the fraction of chains that moved in it (an accepted proposal differs from x)
    _, counts = s.sample(x=x0, n_steps=1, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(5))
    one = s.sample(x=x0, n_steps=1, generator=torch.Generator(device=cuda_device).manual_seed(5))
    moved = (one != x0).any(dim=1).float().mean()
    torch.testing.assert_close(counts["acceptance_rate"][0], moved, rtol=0, atol=1e-6)


def test_c_abi_layout_and_injected_noise_records(cuda_device):
    """The layout query mirrors the dispatch; records with injected noise run on the lane-group kernel."""
    spec = ta.DoubleWellModel(device=cuda_device).fused_spec()
    c = spec.to_c()
    assert _lib.diag_layout(c, _lib.DIAG_LANGEVIN, 1 << 20, 64) == (65536, 64, 1024)       # flat kernel, BASELINE config 2
    assert _lib.diag_layout(c, _lib.DIAG_LANGEVIN, 1000, 100) == (125, 100, 800)           # ragged dim: 8 whole rows per block
    assert _lib.diag_layout(c, _lib.DIAG_LANGEVIN, 10, 4096) == (40, 1024, 1024)           # a row is four blocks
    assert _lib.diag_layout(c, _lib.DIAG_LANGEVIN, 10, 5000) is None                       # neither divides: no in-kernel form
    assert _lib.diag_layout(c, _lib.DIAG_HMC, 10, 2000) is None
    mlp = ta.MLPEnergy(2, device=cuda_device).fused_spec().to_c()
    assert _lib.diag_layout(mlp, _lib.DIAG_LANGEVIN, 100, 2) == (4, 2, 64)                 # MLP kernels: one record per wave of 32 chains
    assert _lib.diag_layout(mlp, _lib.DIAG_LANGEVIN_HEUN, 100, 2) is None                  # no fused Heun kernel for this energy
    n, dim, k = 500, 64, 6
    x0 = torch.randn(n, dim, generator=torch.Generator().manual_seed(0)).clamp_(-2, 2)
    noise = torch.randn(k, n, dim, generator=torch.Generator().manual_seed(1))
    wx, _, wdiag = oracle.langevin_chain(oracle.DoubleWell(), x0, noise, [0.01] * k, [1.0] * k, thin=2, want_diag=True)
    layout = _lib.diag_layout(c, _lib.DIAG_LANGEVIN, n, dim, True, False)
    assert layout == (32, 64, 1024)  # injected noise: lane-group kernel, 16 lanes per chain -> 16 rows per block
    nb, S, E = layout
    rec = torch.empty(3 * nb * (2 * S + 8), device=cuda_device)
    work = torch.zeros(3 * (3 * dim + 3), dtype=torch.float64, device=cuda_device)
    x, nz = x0.to(cuda_device), noise.to(cuda_device)
    st = _lib.stream_handle(cuda_device)
    _lib.call("ebm_langevin_chain_f32", c, x.data_ptr(), n, dim, k, 0.01, 0.01**0.5, 2.0**0.5, None, 0, 0.0, 0.0, 2, None,
              rec.data_ptr(), nz.data_ptr(), 0, 0, st)
    mean, var, en = torch.empty(3, dim, device=cuda_device), torch.empty(3, dim, device=cuda_device), torch.empty(3, device=cuda_device)
    _lib.call("ebm_diag_finish_f32", rec.data_ptr(), 3, nb, S, E, n, dim, mean.data_ptr(), var.data_ptr(), en.data_ptr(), None,
              work.data_ptr(), st)
    assert torch.equal(x.cpu(), wx)
    torch.testing.assert_close(mean.cpu(), wdiag["mean"], rtol=1e-5, atol=2e-6)
    torch.testing.assert_close(var.cpu(), wdiag["var"], rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(en.cpu(), wdiag["energy"], rtol=1e-5, atol=1e-4)
    assert torch.count_nonzero(work) == 0  # the merge leaves its workspace zeroed
    with pytest.raises(ValueError, match="inconsistent layout"):
        _lib.call("ebm_diag_finish_f32", rec.data_ptr(), 3, nb + 1, S, E, n, dim, mean.data_ptr(), var.data_ptr(), en.data_ptr(), None,
                  work.data_ptr(), st)
    with pytest.raises(RuntimeError, match="no in-kernel diagnostics"):
        xw = torch.zeros(4, 5000, device=cuda_device)
        _lib.call("ebm_langevin_chain_f32", c, xw.data_ptr(), 4, 5000, 2, 0.01, 0.1, 1.4, None, 0, 0.0, 0.0, 1, None, rec.data_ptr(),
                  None, 0, 0, st)


def test_config2_with_diagnostics_is_one_launch_at_full_size(cuda_device):
    """BASELINE config 2 with thin = 50 and return_diagnostics=True: one chain launch, and its
    duration stays close to the launch without diagnostics (a generous bound here; profiles/ has the measurement)."""
    n, dim, k = 1 << 20, 64, 200
    model = ta.DoubleWellModel(device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.01, device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device).clamp_(-4.0, 4.0)
    gen = lambda: torch.Generator(device=cuda_device).manual_seed(2)  # noqa: E731
    s.sample(x=x0, n_steps=k, generator=gen())
    c0 = hip_calls("ebm_langevin_chain_f32")
    _lib.timed_events["ebm_langevin_chain_f32"] = []
    fin = s.sample(x=x0, n_steps=k, generator=gen())
    out, diag = s.sample(x=x0, n_steps=k, thin=50, return_diagnostics=True, generator=gen())
    torch.cuda.synchronize()
    pairs = _lib.timed_events.pop("ebm_langevin_chain_f32")
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 2 and len(pairs) == 2
    t_plain, t_diag = pairs[0][0].elapsed_time(pairs[0][1]), pairs[1][0].elapsed_time(pairs[1][1])
    assert t_diag < 1.3 * t_plain, (t_plain, t_diag)  # measured 1.04; the bound only catches a structural regression
    assert torch.equal(fin, out) and diag["mean"].shape == (4, dim)
    xs = out.double()
    torch.testing.assert_close(diag["mean"][3].double(), xs.mean(dim=0), rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(diag["var"][3].double(), xs.var(dim=0, unbiased=False), rtol=1e-4, atol=1e-7)
    torch.testing.assert_close(diag["energy"][3].double(), model(out).double().mean(), rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("kind,dim", [("gauss", 100), ("gauss", 128), ("gmm", 64), ("gmm", 128)])
def test_asking_for_diagnostics_does_not_change_the_chains(cuda_device, kind, dim):
    """records come from the SAME kernel family as the plain call at every dim the matrix-layout kernels take
    (one record per wave of 32 chains, no LDS tile), so sample(return_diagnostics=True) returns the chains of sample()."""
    g = torch.Generator().manual_seed(dim)
    if kind == "gauss":
        a = torch.randn(dim, dim, generator=g)
        model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=cuda_device)
    else:
        model = ta.GaussianMixtureModel(torch.randn(12, dim, generator=g) * 1.5, sigma=1.0, device=cuda_device)
    x0 = torch.randn(1000, dim, device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.01, device=cuda_device)
    plain = s.sample(x=x0, n_steps=12, generator=torch.Generator(device=cuda_device).manual_seed(1))
    with_d, d = s.sample(x=x0, n_steps=12, thin=4, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(1))
    assert torch.equal(plain, with_d)
    torch.testing.assert_close(d["mean"][-1], with_d.mean(dim=0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(d["energy"][-1], model(with_d).mean(), rtol=1e-4, atol=1e-4)
    if kind == "gauss" or dim <= 96:
        h = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=4, device=cuda_device)
        plain = h.sample(x=x0, n_steps=6, generator=torch.Generator(device=cuda_device).manual_seed(2))
        with_d, d = h.sample(x=x0, n_steps=6, thin=2, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(2))
        assert torch.equal(plain, with_d)
        torch.testing.assert_close(d["mean"][-1], with_d.mean(dim=0), rtol=1e-5, atol=1e-6)
