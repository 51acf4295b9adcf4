"""Mixtures whose component means differ only in the first four columns (include/ebm_hip.h: the `aux` active-column
mask of EBM_ENERGY_GMM; csrc/rows.h: Energy<kGmmSlot1>) -- BASELINE config 3's eight-mode ring is one.  The
one-lane-per-chain kernels (dim 16 / 32, K <= 8) then run their K x dim passes over four columns; every other column
is a plain Gaussian coordinate.  Same results as the dense kernel and as the oracle, to the HMC / Langevin tolerance
tiers; accept masks bit-identical."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients
from torchebm_amd.integrators.symplectic import _mass_args

pytestmark = pytest.mark.gpu


def _means(kind, k, dim, seed=0):
    g = torch.Generator().manual_seed(seed)
    m = torch.zeros(k, dim)
    if kind == "ring":      # the config-3 shape: a circle in columns (0, 1), zero elsewhere
        return ta.core.ring_mixture(k, dim).means.cpu().clone()
    if kind == "plane4":    # all four active columns used, non-zero shared columns
        m[:] = torch.randn(dim, generator=g)
        m[:, :4] = torch.randn(k, 4, generator=g) * 2.0
        return m
    if kind == "slot1":     # the differing columns sit in slot 1: the dense kernel must take it
        m[:] = torch.randn(dim, generator=g)
        m[:, 4:6] = torch.randn(k, 2, generator=g) * 2.0
        return m
    raise ValueError(kind)


def _mask_of(model):
    return int(model.fused_spec().aux.item())


@pytest.mark.parametrize("kind,k,dim", [("ring", 8, 32), ("ring", 5, 32), ("plane4", 8, 32), ("plane4", 3, 16), ("slot1", 8, 32)])
@pytest.mark.parametrize("mass", [None, 1.7, "diag"])
def test_hmc_matches_oracle_and_dense_kernel(cuda_device, kind, k, dim, mass):
    means = _means(kind, k, dim)
    model = ta.GaussianMixtureModel(means, sigma=1.0 if kind == "ring" else 0.9, device=cuda_device)
    assert _mask_of(model) == (2 if kind == "slot1" else 1)
    en = oracle.GaussianMixture(means, model.sigma)
    g = torch.Generator().manual_seed(11)
    if mass == "diag":
        mass = torch.rand(dim, generator=g) + 0.5
    n, T, L, eps = 300, 5, 6, 0.08
    x0 = torch.randn(n, dim, generator=g) * 1.5
    p, u = torch.randn(T, n, dim, generator=g), torch.rand(T, n, generator=g)
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, mass=mass, want_traj=True)
    spec = model.fused_spec()

    def run(desc):
        x = x0.to(cuda_device)
        mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
        traj = torch.empty(n, T, dim, device=cuda_device)
        kind_, ms, md = _mass_args(mass.to(cuda_device) if torch.is_tensor(mass) else mass, x)
        p_d, u_d = p.to(cuda_device), u.to(cuda_device)
        _lib.call("ebm_hmc_chain_f32", desc, x.data_ptr(), n, dim, T, L, eps, None, kind_, ms, _lib.ptr(md), 1, traj.data_ptr(), None,
                  mask.data_ptr(), None, p_d.data_ptr(), u_d.data_ptr(), 0, 0, _lib.stream_handle(cuda_device))
        return x.cpu(), mask.cpu().bool(), traj.cpu()

    x_a, m_a, t_a = run(spec.to_c())
    dense = spec.to_c()
    dense.aux = None           # no hint: the dense kernel
    x_b, m_b, t_b = run(dense)
    if want["margin"] > 2e-4:
        assert torch.equal(m_a, want["accepted"]) and torch.equal(m_b, want["accepted"])
        scale = want["trajectory"].abs().clamp(min=1.0)
        assert ((t_a - want["trajectory"]).abs() / scale).max().item() <= 5e-4
        assert ((t_b - want["trajectory"]).abs() / scale).max().item() <= 5e-4
    assert torch.isfinite(x_a).all() and ((x_a - x_b).abs() / x_b.abs().clamp(min=1.0)).max().item() <= 1e-3


@pytest.mark.parametrize("kind,k,dim", [("ring", 8, 32), ("plane4", 6, 32), ("plane4", 8, 16), ("slot1", 4, 16)])
def test_langevin_matches_oracle(cuda_device, kind, k, dim):
    means = _means(kind, k, dim, seed=3)
    model = ta.GaussianMixtureModel(means, sigma=0.8, device=cuda_device)
    en = oracle.GaussianMixture(means, 0.8)
    g = torch.Generator().manual_seed(5)
    n, steps = 500, 12
    x0 = torch.randn(n, dim, generator=g) * 1.5
    noise = torch.randn(steps, n, dim, generator=g)
    want, wtraj, _ = oracle.langevin_chain(en, x0, noise, [0.02] * steps, [1.0] * steps, thin=3, want_traj=True)
    x = x0.to(cuda_device)
    traj = torch.empty(n, steps // 3, dim, device=cuda_device)
    a, sq, coef = em_coefficients(0.02, 1.0)
    nz = noise.to(cuda_device)
    _lib.call("ebm_langevin_chain_f32", model.fused_spec().to_c(), x.data_ptr(), n, dim, steps, a, sq, coef, None, 0, 0.0, 0.0, 3,
              traj.data_ptr(), None, nz.data_ptr(), 0, 0, _lib.stream_handle(cuda_device))
    torch.testing.assert_close(x.cpu(), want, rtol=3e-5, atol=3e-5)
    torch.testing.assert_close(traj.cpu(), wtraj, rtol=3e-5, atol=3e-5)


def test_energy_and_diagnostics_and_safe_mode(cuda_device):
    """Through the public sampler: diagnostics (in-kernel records) agree with torch reductions, extreme states
    stay finite (the shared columns feed the same non-finite check as the dense path), the mask follows the means."""
    model = ta.core.ring_mixture(8, 32, device=cuda_device)
    assert _mask_of(model) == 1
    s = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=8, device=cuda_device)
    x0 = torch.randn(4096, 32, device=cuda_device)
    c0 = hip_calls("ebm_hmc_chain_f32")
    traj, diag = s.sample(x=x0, n_steps=6, thin=2, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(1))
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1 and (diag["acceptance_rate"] > 0.9).all()
    for j in range(3):
        torch.testing.assert_close(diag["energy"][j], model(traj[:, j]).mean(), rtol=1e-4, atol=1e-4)
        torch.testing.assert_close(diag["mean"][j], traj[:, j].mean(dim=0), rtol=1e-4, atol=1e-5)
    wild = x0.clone()
    wild[:7, 5] = 3e38      # overflows in the shared columns
    wild[7:11, 0] = -3e38   # ... and in the active slot
    out = s.sample(x=wild, n_steps=3, generator=torch.Generator(device=cuda_device).manual_seed(2))
    assert torch.isfinite(out[11:]).all()
    with torch.no_grad():
        model.means[:, 9] += torch.arange(8.0, device=cuda_device)  # a second slot becomes active: the mask is recomputed
    assert _mask_of(model) == 0b101
    out2 = s.sample(x=x0, n_steps=2, generator=torch.Generator(device=cuda_device).manual_seed(3))
    assert torch.isfinite(out2).all()


def test_means_written_through_data_do_not_leave_a_stale_hint(cuda_device):
    """``means.data.copy_(..)`` keeps the tensor's storage and version; the active-column hint is recomputed at
    every call, so a ring mixture turned into a dense one runs the dense body (same chains as a freshly built model)."""
    dim, k = 32, 8
    ring = ta.core.ring_mixture(k, dim, device=cuda_device)
    dense_means = torch.randn(k, dim, generator=torch.Generator().manual_seed(3)) * 2.0
    fresh = ta.GaussianMixtureModel(dense_means, sigma=1.0, device=cuda_device)
    x0 = torch.randn(4096, dim, device=cuda_device)
    h = ta.HamiltonianMonteCarlo(ring, step_size=0.1, n_leapfrog_steps=5, device=cuda_device)
    h.sample(x=x0, n_steps=2)                                    # the hint of the ring has been used
    assert int(ring.fused_spec().aux.item()) == 1
    ring.means.data.copy_(dense_means.to(cuda_device))
    assert int(ring.fused_spec().aux.item()) == 0xFF
    a = h.sample(x=x0, n_steps=3, generator=torch.Generator(device=cuda_device).manual_seed(4))
    b = ta.HamiltonianMonteCarlo(fresh, step_size=0.1, n_leapfrog_steps=5, device=cuda_device).sample(
        x=x0, n_steps=3, generator=torch.Generator(device=cuda_device).manual_seed(4))
    assert torch.equal(a, b)


@pytest.mark.parametrize("dim", [4, 8, 16, 20, 32])
def test_active_column_hint_is_one_launch_and_the_tensor_formula(cuda_device, dim):
    """ebm_gmm_active_columns_i32 (ABI 5) against the definition, over random sparsity patterns, a NaN mean (differs from
    everything) and a single component (nothing differs); widths the hint does not cover get none."""
    g = torch.Generator().manual_seed(dim)
    for trial in range(8):
        K = [1, 2, 5, 8, 13, 32, 64, 3][trial]
        means = torch.randn(1, dim, generator=g).repeat(K, 1)
        cols = torch.rand(dim, generator=g) < 0.25
        means[:, cols] += torch.randn(K, int(cols.sum()), generator=g)
        if trial == 7:
            means[1, dim - 1] = float("nan")
        want = 0
        for v in range(dim // 4):
            if (means[:, 4 * v:4 * v + 4] != means[:1, 4 * v:4 * v + 4]).any():
                want |= 1 << v
        model = ta.GaussianMixtureModel(means, sigma=1.0, device=cuda_device)
        c0 = _lib.call_counts["ebm_gmm_active_columns_i32"]
        spec = model.fused_spec()
        assert _lib.call_counts["ebm_gmm_active_columns_i32"] == c0 + 1
        assert spec.aux.dtype == torch.int32 and int(spec.aux.item()) == want, (trial, K, want)
    assert ta.GaussianMixtureModel(torch.randn(4, 36), sigma=1.0, device=cuda_device).fused_spec().aux is None
    assert ta.GaussianMixtureModel(torch.randn(4, 6), sigma=1.0, device=cuda_device).fused_spec().aux is None
