"""Gaussian mixtures at widths that are not a multiple of 4 on SHIFTED rows (csrc/gmm_shift.hip, gmm_hmc_shift.hip; the
layout: tests/test_gauss_shift_gpu.py): Langevin and HMC through the samplers -- native Philox draws -- against the oracle
fed the same field (ebm_noise_fill_f32), over the component-count classes, tile counts, mass forms; records against the
trajectory; one launch per call."""

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


def _mixture(K, dim, device, seed=0):
    g = torch.Generator().manual_seed(100 * K + dim + seed)
    means = torch.randn(K, dim, generator=g) * 1.2
    weights = torch.rand(K, generator=g) + 0.2
    model = ta.GaussianMixtureModel(means, sigma=1.1, weights=weights, device=device)
    return model, oracle.GaussianMixture(means, 1.1, log_weights=model.log_weights.detach().cpu()), g


@pytest.mark.parametrize("K,dim", [(K, d) for K in (3, 8, 9, 16, 32) for d in (21, 30, 33, 50, 65, 99, 125)] +
                         # 129 .. 256 dims: five to eight tiles (csrc/gmm_wide.hip), multiples of 4 and shifted rows
                         [(K, d) for K in (8, 16, 32) for d in (126, 129, 132, 158, 160, 190, 200, 253, 254, 256)] +
                         # more than eight components: the one-tile kernels from 9 / 12 dims (below 20 the lane-group kernels lose to them)
                         [(K, d) for K in (9, 16, 32) for d in (9, 12, 13, 16, 18, 19, 20)] +
                         [(K, d) for K in (3, 8) for d in (17, 18, 19, 29)])  # up to eight components: shifted rows from 17 dims
def test_langevin_against_the_oracle(cuda_device, K, dim):
    model, en, g = _mixture(K, dim, cuda_device)
    n, k, thin = 203, 8, 2
    s = ta.LangevinDynamics(model, step_size=0.05, noise_scale=0.9, device=cuda_device)
    x0 = torch.randn(n, dim, generator=g).clamp_(-2.0, 2.0)
    seed = 900 + K + dim
    c0 = hip_calls("ebm_langevin_chain_f32")
    traj, diag = s.sample(x=x0.to(cuda_device), n_steps=k, thin=thin, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(seed))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
    plain = s.sample(x=x0.to(cuda_device), n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(seed))
    if dim + 3 > 32 or K > 8:  # (one tile with K <= 8: the lane-group kernel, records from its own family)
        assert torch.equal(traj[:, -1], plain)
    noise = _field((n, dim), _rng.kernel_seed(seed), range(k), cuda_device).cpu()
    want, want_traj, _ = oracle.langevin_chain(en, x0, noise, [0.05] * k, [0.9] * k, thin=thin, want_traj=True)
    err = ((traj.cpu() - want_traj).abs() / want_traj.abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
    assert (err <= 5e-4).float().mean().item() >= 0.97 and (err <= 5e-3).all(), err.max().item()
    assert err.median().item() <= 2e-5
    t64 = traj.double()
    torch.testing.assert_close(diag["mean"].double(), t64.mean(dim=0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(diag["var"].double(), t64.var(dim=0, unbiased=False), rtol=1e-4, atol=1e-7)
    want_e = torch.stack([model(traj[:, j]).double().mean() for j in range(k // thin)])
    torch.testing.assert_close(diag["energy"].double(), want_e, rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("K", [3, 9, 16, 32])
@pytest.mark.parametrize("dim,mass", [(21, None), (30, 1.7), (33, "diag"), (50, None), (65, "diag"), (93, 0.6), (99, None), (125, 1.3),
                                      (9, None), (12, 1.7), (13, "diag"), (16, None), (18, "diag"), (19, 0.6)])
def test_hmc_against_the_oracle(cuda_device, K, dim, mass):
    model, en, g = _mixture(K, dim, cuda_device, seed=1)
    if mass == "diag":
        mass = torch.rand(dim, generator=g) + 0.5
    n, T, L, thin, eps = 97, 4, 6, 2, 0.12
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L,
                                 mass=mass.to(cuda_device) if torch.is_tensor(mass) else mass, device=cuda_device)
    x0 = torch.randn(n, dim, generator=g).clamp_(-2.0, 2.0)
    seed = 7000 + K + dim
    c0 = hip_calls("ebm_hmc_chain_f32")
    traj = s.sample(x=x0.to(cuda_device), n_steps=T, thin=thin, return_trajectory=True,
                    generator=torch.Generator(device=cuda_device).manual_seed(seed))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    p = _field((n, dim), _rng.kernel_seed(seed), range(0, 2 * T, 2), cuda_device).cpu()
    u = _field((n,), _rng.kernel_seed(seed), range(1, 2 * T, 2), cuda_device, kind=_lib.NOISE_UNIFORM).cpu()
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, mass=mass, thin=thin, want_traj=True, want_diag=True)
    err = ((traj.cpu() - want["trajectory"]).abs() / want["trajectory"].abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
    if want["margin"] > 1e-4:
        assert (err <= 5e-4).float().mean().item() >= 0.97 and (err <= 5e-3).all(), err.max().item()
        assert err.median().item() <= 2e-5
    else:
        assert (err <= 5e-4).float().mean().item() >= 0.9
    # with records (three tiles at most): same chains, the diagnostics of the oracle
    if dim + 3 <= 96 and want["margin"] > 1e-4:
        traj2, diag = s.sample(x=x0.to(cuda_device), n_steps=T, thin=thin, return_trajectory=True, return_diagnostics=True,
                               generator=torch.Generator(device=cuda_device).manual_seed(seed))
        if dim + 3 > 32 or K > 8:
            assert torch.equal(traj2, traj)
        torch.testing.assert_close(diag["acceptance_rate"].cpu(), want["diagnostics"]["acceptance_rate"], rtol=0, atol=1e-6)
        torch.testing.assert_close(diag["mean"].cpu(), want["diagnostics"]["mean"], rtol=1e-3, atol=1e-3)


# 129 .. 256 dims: five to eight tiles of the transition body (csrc/gmm_hmc_wide.hip; widths off multiples of 4 on shifted rows,
# gmm_hmc_wide_shift.hip) -- no mass vector there (the lane-group kernels keep it)
@pytest.mark.parametrize("K", [8, 16, 32])
@pytest.mark.parametrize("dim,mass", [(126, None), (129, 1.7), (132, None), (158, 0.6), (160, None), (190, None), (200, 1.3), (224, None),
                                      (253, None), (254, 0.8), (256, None), (200, "diag")])
def test_wide_hmc_against_the_oracle(cuda_device, K, dim, mass):
    model, en, g = _mixture(K, dim, cuda_device, seed=2)
    if mass == "diag":
        mass = torch.rand(dim, generator=g) + 0.5
    n, T, L, thin, eps = 161, 4, 5, 2, 0.08
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L,
                                 mass=mass.to(cuda_device) if torch.is_tensor(mass) else mass, device=cuda_device)
    x0 = torch.randn(n, dim, generator=g).clamp_(-2.0, 2.0)
    seed = 8000 + K + dim
    c0 = hip_calls("ebm_hmc_chain_f32")
    traj, acc = s.sample(x=x0.to(cuda_device), n_steps=T, thin=thin, return_trajectory=True, return_diagnostics=False,
                         generator=torch.Generator(device=cuda_device).manual_seed(seed)), None
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    p = _field((n, dim), _rng.kernel_seed(seed), range(0, 2 * T, 2), cuda_device).cpu()
    u = _field((n,), _rng.kernel_seed(seed), range(1, 2 * T, 2), cuda_device, kind=_lib.NOISE_UNIFORM).cpu()
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, mass=mass, thin=thin, want_traj=True)
    err = ((traj.cpu() - want["trajectory"]).abs() / want["trajectory"].abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
    if want["margin"] > 1e-4:
        assert (err <= 5e-4).float().mean().item() >= 0.97 and (err <= 5e-3).all(), err.max().item()
        assert err.median().item() <= 2e-5
    else:
        assert (err <= 5e-4).float().mean().item() >= 0.9


@pytest.mark.parametrize("dim", [132, 200, 254, 256])
def test_wide_hmc_injected_draws_give_the_oracles_accept_mask(cuda_device, dim):
    """Through the C ABI with injected momenta and uniforms: the accept mask is the oracle's, bit for bit (seeds without a
    borderline decision), and a NaN chain stays where it is without touching its neighbours."""
    K, n, T, L, eps = 16, 130, 3, 4, 0.1
    model, en, g = _mixture(K, dim, cuda_device, seed=3)
    x0 = torch.randn(n, dim, generator=g).clamp_(-2.0, 2.0)
    x0[7, 3] = float("nan")
    p, u = torch.randn(T, n, dim, generator=g), torch.rand(T, n, generator=g)
    ref = oracle.hmc_chain(en, x0, p, u, [eps] * T, L)
    xd, p_d, u_d = x0.to(cuda_device), p.to(cuda_device), u.to(cuda_device)
    mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
    spec = model.fused_spec()
    _lib.call("ebm_hmc_chain_f32", spec.to_c(), xd.data_ptr(), n, dim, T, L, eps, None, 0, 0.0, None, 1, None, None,
              mask.data_ptr(), None, p_d.data_ptr(), u_d.data_ptr(), 0, 0, _lib.stream_handle(cuda_device))
    got = xd.cpu()
    if ref["margin"] > 1e-4:
        assert torch.equal(mask.cpu().bool(), ref["accepted"])
    assert not mask[:, 7].any() and torch.isnan(got[7, 3])
    keep = torch.ones(n, dtype=torch.bool); keep[7] = False
    err = ((got[keep] - ref["x"][keep]).abs() / ref["x"][keep].abs().clamp(min=1.0)).amax(dim=1)
    assert (err <= 5e-4).float().mean().item() >= 0.97 and (err <= 5e-3).all(), err.max().item()


@pytest.mark.parametrize("K,dim,thin", [(8, 160, 1), (16, 224, 2), (8, 240, 2), (32, 252, 1), (16, 254, 2)])
def test_wide_hmc_mid_call_hand_over_to_the_literal_body(cuda_device, K, dim, thin):
    """(round 6) The wide mixture kernels take the force in pieces with merged kicks (mfma_hmc_body.h PW; GmmE::eval_tiles) and a
    WAVE that meets a non-finite energy or momentum finishes its call -- from the transition at hand, the thinning counter resumed --
    in the literal body (gauss_hmc_fallback).  Absurd momentum draws in transitions 2 and 4: decisions, final states and kept rows of
    all chains are the oracle's (samplers/hmc.py:243-312, integrators/leapfrog.py:165-185)."""
    n, T, L, eps = 200, 6, 4, 0.08
    model, en, g = _mixture(K, dim, cuda_device, seed=4)
    x0 = torch.randn(n, dim, generator=g).clamp_(-2.0, 2.0)
    p, u = torch.randn(T, n, dim, generator=g), torch.rand(T, n, generator=g)
    p[2, 5] *= 1e37
    p[4, 190] *= 1e25
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, thin=thin, want_traj=True, want_margins=True)
    x = x0.to(cuda_device)
    mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
    cnt = torch.zeros(T, dtype=torch.int32, device=cuda_device)
    traj = torch.full((n, T // thin, dim), float("nan"), device=cuda_device)
    pd, ud = p.to(cuda_device), u.to(cuda_device)
    spec = model.fused_spec()
    _lib.call("ebm_hmc_chain_f32", spec.to_c(), x.data_ptr(), n, dim, T, L, eps, None, 0, 0.0, None, thin, traj.data_ptr(), None,
              mask.data_ptr(), cnt.data_ptr(), pd.data_ptr(), ud.data_ptr(), 0, 0, _lib.stream_handle(cuda_device))
    torch.cuda.synchronize()
    x, mask, traj = x.cpu(), mask.cpu().bool(), traj.cpu()
    # The two absurd chains themselves are outside what is compared: at |x| ~ 1e36 every logit of the reference's logsumexp is -inf,
    # its autograd force NaN (scrubbed: p = 0, H1 = 1e10, accepted against H0 = inf), where the kernel's analytic force -- the |x|^2
    # term cancelled -- is finite and its H0 - H1 = inf - inf is rejected.  What the test is about is everybody ELSE in their waves.
    absurd = torch.zeros(n, dtype=torch.bool)
    absurd[[5, 190]] = True
    clear = (want["margins"] > 2e-4) & ~absurd
    assert torch.equal(mask[clear], want["accepted"][clear])
    assert torch.equal(cnt.cpu(), mask.sum(dim=1).to(torch.int32))
    ok = (want["margins"] > 2e-4).all(dim=0) & ~absurd
    assert ok.float().mean().item() > 0.9
    assert torch.isfinite(x[~absurd]).all() and torch.isfinite(traj[~absurd]).all()
    err = ((x[ok] - want["x"][ok]).abs() / want["x"][ok].abs().clamp(min=1.0)).amax(dim=1)
    assert (err <= 5e-4).float().mean().item() >= 0.97 and (err <= 5e-3).all(), err.max().item()
    wt = want["trajectory"]
    errt = ((traj[ok] - wt[ok]).abs() / wt[ok].abs().clamp(min=1.0)).reshape(int(ok.sum()), -1).amax(dim=1)
    assert (errt <= 5e-4).float().mean().item() >= 0.97 and (errt <= 5e-3).all(), errt.max().item()
