"""Distributional parity of the native (in-kernel Philox) path -- SURVEY.md section 7 "KS on marginals".  The CPU reference
draws from mt19937, so native-RNG runs cannot match it draw for draw; what must match is the LAW: Kolmogorov-Smirnov
tests of the kernel's normals / uniforms against N(0,1) / U[0,1), and two-sample KS tests of the stationary marginals
of GPU chains against the oracle's CPU chains (same energy, step size and number of steps, independent noise).
Fixed seeds: the p-values below are reproducible numbers, the bar (p > 1e-3 on every marginal) is written here."""

import numpy as np
import pytest
import torch
from scipy import stats

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib

pytestmark = pytest.mark.gpu
P_MIN = 1e-3


def test_kernel_normals_and_uniforms_follow_their_laws(cuda_device):
    n = 1 << 20
    buf = torch.empty(n, device=cuda_device)
    for step in (0, 12345):
        _lib.call("ebm_noise_fill_f32", buf.data_ptr(), n, _lib.NOISE_NORMAL, 99, step, _lib.stream_handle(cuda_device))
        z = buf.cpu().double().numpy()
        assert stats.kstest(z, "norm").pvalue > P_MIN
        assert abs(z.mean()) < 4 / np.sqrt(n) and abs(z.var() - 1) < 6 * np.sqrt(2 / n)
        assert abs(stats.skew(z)) < 0.02 and abs(stats.kurtosis(z)) < 0.04
        # tails: P(|z| > 4) = 6.33e-5 -> ~66 of 2^20
        assert 30 < (np.abs(z) > 4).sum() < 110
        _lib.call("ebm_noise_fill_f32", buf.data_ptr(), n, _lib.NOISE_UNIFORM, 99, step, _lib.stream_handle(cuda_device))
        u = buf.cpu().double().numpy()
        assert stats.kstest(u, "uniform").pvalue > P_MIN and u.min() >= 0.0 and u.max() < 1.0
    # consecutive elements and consecutive steps are uncorrelated
    _lib.call("ebm_noise_fill_f32", buf.data_ptr(), n, _lib.NOISE_NORMAL, 99, 7, _lib.stream_handle(cuda_device))
    a = buf.cpu().double().numpy()
    _lib.call("ebm_noise_fill_f32", buf.data_ptr(), n, _lib.NOISE_NORMAL, 99, 8, _lib.stream_handle(cuda_device))
    b = buf.cpu().double().numpy()
    assert abs(np.corrcoef(a[:-1], a[1:])[0, 1]) < 5 / np.sqrt(n) and abs(np.corrcoef(a, b)[0, 1]) < 5 / np.sqrt(n)


def _cpu_langevin(energy, n, dim, k, eta, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, dim, generator=g)
    noise = torch.randn(k, n, dim, generator=g)
    out, _, _ = oracle.langevin_chain(energy, x, noise, [eta] * k, [1.0] * k)
    return out


@pytest.mark.parametrize("kind", ["gauss2d", "double_well", "gmm"])
def test_langevin_stationary_marginals_match_the_oracle(cuda_device, kind):
    n, k = 8192, 300
    if kind == "gauss2d":
        mean, cov = torch.tensor([0.5, -0.25]), torch.tensor([[1.0, 0.8], [0.8, 1.0]])
        model, en, dim, eta = ta.GaussianModel(mean, cov, device=cuda_device), oracle.Gaussian(mean, cov), 2, 0.05
    elif kind == "double_well":
        model, en, dim, eta = ta.DoubleWellModel(device=cuda_device), oracle.DoubleWell(), 4, 0.01
    else:
        means = ta.core.ring_mixture(8, 16).means.cpu() * 0.5  # radius 2: the modes mix within 300 steps
        model, en, dim, eta = ta.GaussianMixtureModel(means, device=cuda_device), oracle.GaussianMixture(means, 1.0), 16, 0.05
    want = _cpu_langevin(en, n, dim, k, eta, seed=1).double().numpy()
    s = ta.LangevinDynamics(model, step_size=eta, device=cuda_device)
    gen = torch.Generator(device=cuda_device).manual_seed(2)
    got = s.sample(dim=dim, n_samples=n, n_steps=k, generator=gen).cpu().double().numpy()
    for c in range(min(dim, 4)):
        p = stats.ks_2samp(got[:, c], want[:, c]).pvalue
        assert p > P_MIN, (kind, c, p)
    if kind == "gauss2d":  # ... and against the closed-form law of the discretised chain: x' = x - eta P (x - mu) + sqrt(2 eta) eps
        prec = np.linalg.inv(cov.double().numpy())
        w, v = np.linalg.eigh(prec)
        var_modes = 1.0 / (w * (1.0 - eta * w / 2.0))          # stationary variance of each eigen-direction
        proj = (got - mean.double().numpy()) @ v
        for c in range(2):
            assert stats.kstest(proj[:, c] / np.sqrt(var_modes[c]), "norm").pvalue > P_MIN


def test_hmc_marginals_and_acceptance_match_the_oracle(cuda_device):
    n, T, L, eps = 4096, 30, 8, 0.2
    mean, cov = torch.tensor([0.5, -0.25]), torch.tensor([[1.0, 0.8], [0.8, 1.0]])
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(n, 2, generator=g)
    p, u = torch.randn(T, n, 2, generator=g), torch.rand(T, n, generator=g)
    want = oracle.hmc_chain(oracle.Gaussian(mean, cov), x0, p, u, [eps] * T, L, want_diag=True)
    s = ta.HamiltonianMonteCarlo(ta.GaussianModel(mean, cov, device=cuda_device), step_size=eps, n_leapfrog_steps=L, device=cuda_device)
    got, diag = s.sample(x=x0.to(cuda_device), n_steps=T, return_diagnostics=True, generator=torch.Generator(device=cuda_device).manual_seed(4))
    gx, wx = got.cpu().double().numpy(), want["x"].double().numpy()
    for c in range(2):
        assert stats.ks_2samp(gx[:, c], wx[:, c]).pvalue > P_MIN
        # HMC is exact: the target's own marginal N(mean_c, cov_cc)
        assert stats.kstest((gx[:, c] - mean[c].item()) / np.sqrt(cov[c, c].item()), "norm").pvalue > P_MIN
    acc_gpu, acc_cpu = diag["acceptance_rate"].mean().item(), want["diagnostics"]["acceptance_rate"].mean().item()
    assert abs(acc_gpu - acc_cpu) < 0.01, (acc_gpu, acc_cpu)


def test_matrix_pipe_kernels_sample_the_target_law(cuda_device):
    """The kernels that contract on the bf16 pipe with split operands (dense Gaussian, dim 64; a 16-component mixture,
    dim 32): HMC is exact, so after burn-in every marginal must follow the TARGET's own marginal -- N(mu_c, Sigma_cc) for
    the Gaussian, the 1-D mixture sum_k w_k N(mu_kc, sigma^2) for the mixture -- and Langevin with a small step must agree
    with the oracle's CPU chain of the same length."""
    n = 8192
    # dense Gaussian, dim 64
    g = torch.Generator().manual_seed(64)
    dim = 64
    a = torch.randn(dim, dim, generator=g)
    cov = a @ a.t() / dim + 0.5 * torch.eye(dim)
    mean = torch.randn(dim, generator=g)
    s = ta.HamiltonianMonteCarlo(ta.GaussianModel(mean, cov, device=cuda_device), step_size=0.15, n_leapfrog_steps=10, device=cuda_device)
    x = s.sample(x=torch.randn(n, dim, generator=g).to(cuda_device), n_steps=60,
                 generator=torch.Generator(device=cuda_device).manual_seed(1)).cpu().double().numpy()
    for c in (0, 17, 40, 63):
        assert stats.kstest((x[:, c] - mean[c].item()) / np.sqrt(cov[c, c].item()), "norm").pvalue > P_MIN, c
    emp = np.cov(x, rowvar=False)
    assert np.abs(emp - cov.double().numpy()).max() < 0.12
    # 16-component mixture, dim 32, well-mixed modes (sigma comparable to the spread of the means)
    g = torch.Generator().manual_seed(16)
    K, dim, sigma = 16, 32, 1.0
    means = torch.randn(K, dim, generator=g) * 0.8
    w = torch.rand(K, generator=g) + 0.5
    model = ta.GaussianMixtureModel(means, sigma=sigma, weights=w, device=cuda_device)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.2, n_leapfrog_steps=12, device=cuda_device)
    x = s.sample(x=torch.randn(n, dim, generator=g).to(cuda_device), n_steps=150,
                 generator=torch.Generator(device=cuda_device).manual_seed(2)).cpu().double().numpy()
    wn = (w / w.sum()).double().numpy()
    for c in (0, 9, 31):
        mu_c = means[:, c].double().numpy()
        cdf = lambda t, mu_c=mu_c: sum(wk * stats.norm.cdf(t, loc=m, scale=sigma) for wk, m in zip(wn, mu_c))  # noqa: E731
        assert stats.kstest(x[:, c], cdf).pvalue > P_MIN, c
    # Langevin on the same mixture against the oracle's chain (independent noise, same law)
    k, eta = 400, 0.02
    ld = ta.LangevinDynamics(model, step_size=eta, device=cuda_device)
    x0 = torch.randn(2048, dim, generator=g)
    got = ld.sample(x=x0.to(cuda_device), n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(3)).cpu().double().numpy()
    en = oracle.GaussianMixture(means, sigma, log_weights=model.log_weights.detach().cpu())
    xc = x0.clone()
    gn = torch.Generator().manual_seed(5)
    for _ in range(k):
        xc = oracle.em_step(xc, en.grad_autograd(xc), torch.randn(2048, dim, generator=gn), eta, 1.0)
    ref = xc.double().numpy()
    for c in (0, 9, 31):
        assert stats.ks_2samp(got[:, c], ref[:, c]).pvalue > P_MIN, c


def test_config3_native_rng_acceptance_and_marginals_match_the_oracle(cuda_device):
    """VERDICT r4 item 4 / BASELINE.md section 5: BASELINE config 3's OWN energy (8-mode ring mixture, dim 32, L = 20, eps = 0.1)
    through `sample()` with the in-kernel draws (n = 2^16) against the oracle's CPU chain with independent mt19937 draws
    (n = 2^14): the acceptance rate of every transition within 4 sigma of the binomial error of the two populations, and the
    radial and per-column marginals after the last transition by two-sample Kolmogorov-Smirnov.  Tolerance source:
    torchebm/tests/samplers/test_hmc.py:668-703 (statistical recovery), samplers/hmc.py:243-312 (the transition)."""
    n_gpu, n_cpu, dim, T, L, eps = 1 << 16, 1 << 14, 32, 10, 20, 0.1
    model = ta.core.ring_mixture(8, dim, device=cuda_device)
    means = model.means.detach().cpu()
    g = torch.Generator().manual_seed(33)
    x0 = torch.randn(n_cpu, dim, generator=g)
    p, u = torch.randn(T, n_cpu, dim, generator=g), torch.rand(T, n_cpu, generator=g)
    want = oracle.hmc_chain(oracle.GaussianMixture(means, 1.0), x0, p, u, [eps] * T, L, want_diag=True)
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, device=cuda_device)
    # (fixed seeds on both sides: the p-values below are deterministic; over three GPU seeds the smallest two-sample p of all 32
    #  columns was 3.5e-4 / 0.015 / 0.007 -- the first an unlucky pairing with the CPU sample's own column 2, p = 0.006 against N(0, 1))
    gen = torch.Generator(device=cuda_device).manual_seed(35)
    start = torch.randn(n_gpu, dim, device=cuda_device, generator=gen)
    c0 = hip_calls("ebm_hmc_chain_f32")
    got, diag = s.sample(x=start, n_steps=T, return_diagnostics=True, generator=gen)
    assert hip_calls("ebm_hmc_chain_f32") > c0  # the fused route (config 3's own kernels), not per-transition launches
    acc_gpu = diag["acceptance_rate"].double().cpu().numpy()
    acc_cpu = want["accepted"].double().mean(dim=1).numpy()
    assert acc_gpu.shape == acc_cpu.shape == (T,)
    for t in range(T):
        pbar = (acc_gpu[t] * n_gpu + acc_cpu[t] * n_cpu) / (n_gpu + n_cpu)
        sigma = np.sqrt(max(pbar * (1.0 - pbar), 1e-6) * (1.0 / n_gpu + 1.0 / n_cpu))
        assert abs(acc_gpu[t] - acc_cpu[t]) < 4.0 * sigma + 1e-4, (t, acc_gpu[t], acc_cpu[t], sigma)
    gx, wx = got.cpu().double().numpy(), want["x"].double().numpy()
    assert np.isfinite(gx).all()
    assert stats.ks_2samp(np.hypot(gx[:, 0], gx[:, 1]), np.hypot(wx[:, 0], wx[:, 1])).pvalue > P_MIN  # distance from the ring's axis
    for c in (0, 1, 2, 17, 31):
        assert stats.ks_2samp(gx[:, c], wx[:, c]).pvalue > P_MIN, c
    # the inactive columns are exact unit Gaussians under the target; 10 transitions of trajectory length 2 from N(0, 1) keep them so
    assert stats.kstest(gx[:, 20], "norm").pvalue > P_MIN
