"""The HMC kernel of its own for mixtures whose means differ in columns 0..3 only (csrc/hmc_ring.hip; BASELINE config 3's
ring): four waves per SIMD, shared columns held as x - mu_0, safe mode through running maxima and a literal redo.

Checked here, through the C ABI: the kernel's own Philox draws are the field ebm_noise_fill_f32 materialises; chain
counts off the workgroup size; per-transition step sizes; records on / off give the same chains; the literal safe-mode
sequence (forces beyond the clamp, non-finite coordinates) against the oracle's reference semantics; the two bodies
(two / four active columns) against the oracle and the general kernel."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from torchebm_amd import _lib

pytestmark = pytest.mark.gpu


def _plane_means(k, dim, cols, seed=0, shared_zero=False):
    """k means that differ in `cols` (inside ONE aligned block of four columns) only; the shared columns random (or zero)."""
    g = torch.Generator().manual_seed(seed)
    m = torch.zeros(k, dim)
    if not shared_zero:
        m[:] = torch.randn(dim, generator=g)
    for c in cols:
        m[:, c] = torch.randn(k, generator=g) * 2.5
    return m


def _run(desc, x0, T, L, eps, dev, p=None, u=None, seed=0, offset=0, thin=1, want_traj=False, table=None, records=None):
    n, dim = x0.shape
    x = x0.to(dev)
    mask = torch.empty(T, n, dtype=torch.uint8, device=dev)
    cnt = torch.zeros(T, dtype=torch.int32, device=dev)
    traj = torch.empty(n, T // thin, dim, device=dev) if want_traj else None
    p_d = p.to(dev) if p is not None else None
    u_d = u.to(dev) if u is not None else None
    tab = torch.tensor(table, dtype=torch.float32, device=dev) if table is not None else None
    _lib.call("ebm_hmc_chain_f32", desc, x.data_ptr(), n, dim, T, L, eps, _lib.ptr(tab), 0, 0.0, None, thin, _lib.ptr(traj),
              _lib.ptr(records), mask.data_ptr(), cnt.data_ptr(), _lib.ptr(p_d), _lib.ptr(u_d), seed, offset, _lib.stream_handle(dev))
    torch.cuda.synchronize()
    return x.cpu(), mask.cpu().bool(), cnt.cpu(), (traj.cpu() if want_traj else None)


@pytest.mark.parametrize("cols,k", [((0, 1), 8), ((0, 1, 2, 3), 8), ((0, 3), 5), ((1,), 3),
                                    ((8, 9), 8), ((20, 21, 22, 23), 6), ((28, 31), 4)])  # ... and in another slot (round 4)
@pytest.mark.parametrize("n", [1, 255, 700])
def test_bodies_match_oracle_and_general_kernel(cuda_device, cols, k, n):
    dim, T, L, eps = 32, 6, 7, 0.09
    means = _plane_means(k, dim, cols, seed=len(cols) + k)
    model = ta.GaussianMixtureModel(means, sigma=0.9, device=cuda_device)
    assert int(model.fused_spec().aux.item()) == 1 << (cols[0] // 4)
    en = oracle.GaussianMixture(means, 0.9)
    g = torch.Generator().manual_seed(n)
    x0 = torch.randn(n, dim, generator=g) * 1.5
    p, u = torch.randn(T, n, dim, generator=g), torch.rand(T, n, generator=g)
    table = [eps * (1.0 + 0.1 * t) for t in range(T)]
    want = oracle.hmc_chain(en, x0, p, u, table, L, thin=2, want_traj=True, want_margins=True)
    spec = model.fused_spec()  # (kept alive: the descriptor points into its tensors)
    desc = spec.to_c()
    x, mask, cnt, traj = _run(desc, x0, T, L, eps, cuda_device, p=p, u=u, thin=2, want_traj=True, table=table)
    general = spec.to_c()
    general.aux = None
    xg, mg, _, _ = _run(general, x0, T, L, eps, cuda_device, p=p, u=u, thin=2, want_traj=True, table=table)
    clear = want["margins"] > 2e-4                       # decisions that are not a rounding away from flipping
    assert torch.equal(mask[clear], want["accepted"][clear]) and torch.equal(mg[clear], want["accepted"][clear])
    assert torch.equal(cnt, mask.sum(dim=1).to(torch.int32))
    if bool(clear.all()):
        scale = want["trajectory"].abs().clamp(min=1.0)
        assert ((traj - want["trajectory"]).abs() / scale).max().item() <= 5e-4
        assert ((x - want["x"]).abs() / want["x"].abs().clamp(min=1.0)).max().item() <= 5e-4
        assert ((x - xg).abs() / xg.abs().clamp(min=1.0)).max().item() <= 1e-3


def test_native_draws_are_the_materialised_field(cuda_device):
    n, dim, T, L, eps, seed, step0 = 1000, 32, 5, 6, 0.1, 77, 40
    model = ta.core.ring_mixture(8, dim, device=cuda_device)
    spec = model.fused_spec()
    desc = spec.to_c()
    x0 = torch.randn(n, dim, generator=torch.Generator().manual_seed(1))
    p = torch.empty(T, n, dim, device=cuda_device)
    u = torch.empty(T, n, device=cuda_device)
    st = _lib.stream_handle(cuda_device)
    for t in range(T):
        _lib.call("ebm_noise_fill_f32", p[t].data_ptr(), n * dim, _lib.NOISE_NORMAL, seed, step0 + 2 * t, st)
        _lib.call("ebm_noise_fill_f32", u[t].data_ptr(), n, _lib.NOISE_UNIFORM, seed, step0 + 2 * t + 1, st)
    a = _run(desc, x0, T, L, eps, cuda_device, seed=seed, offset=step0, want_traj=True)
    b = _run(desc, x0, T, L, eps, cuda_device, p=p.cpu(), u=u.cpu(), want_traj=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and torch.equal(a[3], b[3])
    assert 0.9 < a[1].float().mean().item() <= 1.0


@pytest.mark.parametrize("cols", [(0, 1), (0, 1, 2, 3)])
def test_records_do_not_change_the_chains(cuda_device, cols):
    n, dim, T, L, eps = 777, 32, 6, 5, 0.1
    means = _plane_means(8, dim, cols, seed=3)
    model = ta.GaussianMixtureModel(means, sigma=1.0, device=cuda_device)
    s = ta.HamiltonianMonteCarlo(model, step_size=eps, n_leapfrog_steps=L, device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device)
    traj, d = s.sample(x=x0, n_steps=T, thin=2, return_trajectory=True, return_diagnostics=True,
                       generator=torch.Generator(device=cuda_device).manual_seed(9))
    traj2 = s.sample(x=x0, n_steps=T, thin=2, return_trajectory=True, generator=torch.Generator(device=cuda_device).manual_seed(9))
    assert torch.equal(traj, traj2)
    for j in range(T // 2):
        torch.testing.assert_close(d["mean"][j], traj[:, j].mean(dim=0), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(d["var"][j], traj[:, j].var(dim=0, unbiased=False), rtol=1e-3, atol=1e-5)
        torch.testing.assert_close(d["energy"][j], model(traj[:, j]).mean(), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("cols,shared_zero", [((0, 1), True), ((0, 1, 2, 3), False)])
def test_safe_mode_literal_sequence_matches_oracle(cuda_device, cols, shared_zero):
    """Chains whose forces leave the +-1e6 clamp, or whose coordinates are not finite, are redone by the literal sequence
    (leapfrog.py:165-185: NaN-propagating clamp, half kicks, nan_to_num_ after every step).  Known answers: the oracle."""
    n, dim, T, L, eps = 256, 32, 4, 5, 0.05
    means = _plane_means(8, dim, cols, seed=11, shared_zero=shared_zero)
    model = ta.GaussianMixtureModel(means, sigma=1.0, device=cuda_device)
    en = oracle.GaussianMixture(means, 1.0)
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(n, dim, generator=g)
    x0[0, 9] = 5e6            # shared column: force -5e6, clamped
    x0[1, 0] = -3e6           # active column: clamped
    x0[2, 17] = 3e38          # overflows within the first step
    x0[3, 1] = float("inf")
    x0[4, 30] = float("nan")
    x0[5, 0] = float("nan")
    x0[6, 12] = -2e7
    x0[6, 13] = 4e6
    p, u = torch.randn(T, n, dim, generator=g), torch.rand(T, n, generator=g)
    p[0, 7, 3] = 3e38         # a momentum outside the tame range
    p[1, 8, 20] = float("inf")
    want = oracle.hmc_chain(en, x0, p, u, [eps] * T, L, want_margins=True)
    spec = model.fused_spec()
    assert int(spec.aux.item()) == 1
    x, mask, _, _ = _run(spec.to_c(), x0, T, L, eps, cuda_device, p=p, u=u)
    clear = want["margins"] > 2e-4
    assert torch.equal(mask[clear], want["accepted"][clear])
    tame = torch.ones(n, dtype=torch.bool)
    tame[:9] = False
    assert torch.isfinite(x[tame]).all()
    assert ((x[tame] - want["x"][tame]).abs() / want["x"][tame].abs().clamp(min=1.0)).max().item() <= 5e-4
    # the wild chains: same non-finite pattern, same values to a relative 1e-3 where the oracle's are finite
    wild_got, wild_want = x[:9], want["x"][:9]
    assert torch.equal(torch.isnan(wild_got), torch.isnan(wild_want))
    fin = torch.isfinite(wild_want)
    assert torch.equal(torch.isfinite(wild_got), fin)
    assert ((wild_got[fin] - wild_want[fin]).abs() / wild_want[fin].abs().clamp(min=1.0)).max().item() <= 1e-3
