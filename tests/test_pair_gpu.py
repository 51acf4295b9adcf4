"""dim == 2, Gaussian / Gaussian-mixture energies: the two-chains-per-lane Langevin kernel
(csrc/rows_langevin.hip: langevin_chain_pair_kernel) against the oracle and against itself.

* in-kernel Philox draws == the same field materialised by ebm_noise_fill_f32 and injected (bit for bit):
  the pair layout consumes one counter per two chains, the field is addressed by flat element all the same;
* injected field vs the oracle's chain on the CPU (Gaussian / mixture tolerance of tests/test_langevin_gpu.py);
* odd chain counts (a lane with one chain), a single chain, partial workgroups, a state view that is only 8-byte
  aligned (through the sampler), clamp + schedule + thinning + trajectory, Heun, and the diagnostics records."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from torchebm_amd import _lib, _rng
from torchebm_amd.samplers.langevin import em_coefficients

pytestmark = pytest.mark.gpu


def _models(kind, device):
    g = torch.Generator().manual_seed(5)
    if kind == "gauss":
        a = torch.randn(2, 2, generator=g)
        mean, cov = torch.tensor([0.4, -0.7]), a @ a.t() / 2 + 0.5 * torch.eye(2)
        return ta.GaussianModel(mean, cov, device=device), oracle.Gaussian(mean, cov)
    k = {"gmm3": 3, "gmm8": 8, "gmm11": 11}[kind]
    means = torch.randn(k, 2, generator=g) * 1.5
    return ta.GaussianMixtureModel(means, sigma=0.9, device=device), oracle.GaussianMixture(means, 0.9)


def _field(n, seed, k, device):
    rows = []
    for s in range(k):  # (one allocation per step: a slice of a [k, n, 2] buffer is not 16-byte aligned for odd n)
        buf = torch.empty(n, 2, device=device)
        _lib.call("ebm_noise_fill_f32", buf.data_ptr(), 2 * n, _lib.NOISE_NORMAL, seed, s, _lib.stream_handle(device))
        rows.append(buf)
    return torch.stack(rows)


def _call(spec, x, k, rows, clamp, thin, traj, noise, seed, entry="ebm_langevin_chain_f32", records=None):
    n = x.shape[0]
    a, sq, coef = rows[0]
    table = None
    if len(rows) > 1:
        table = torch.tensor([(r[0], r[1], r[2], 0.0) for r in rows], dtype=torch.float32, device=x.device)
    con, cmin, cmax = (0, 0.0, 0.0) if clamp is None else (1, clamp[0], clamp[1])
    _lib.call(entry, spec.to_c(), x.data_ptr(), n, 2, k, a, sq, coef, _lib.ptr(table), con, cmin, cmax, thin,
              _lib.ptr(traj), _lib.ptr(records), _lib.ptr(noise), seed, 0, _lib.stream_handle(x.device))
    return table


@pytest.mark.parametrize("kind", ["gauss", "gmm3", "gmm8", "gmm11"])
@pytest.mark.parametrize("n", [1, 2, 777, 4099])
@pytest.mark.parametrize("variant", ["plain", "sched_clamp_thin", "heun"])
def test_pair_kernel_native_equals_injected_equals_oracle(cuda_device, kind, n, variant):
    model, en = _models(kind, cuda_device)
    spec = model.fused_spec()
    k = 12
    x0 = torch.randn(n, 2, generator=torch.Generator().manual_seed(n)).clamp_(-2, 2)
    etas, sigs, clamp, thin = [0.02] * k, [1.0] * k, None, 1
    if variant == "sched_clamp_thin":
        etas = [0.03 * 0.9 ** i for i in range(k)]
        sigs = [1.0 - 0.03 * i for i in range(k)]
        clamp, thin = (-1.5, 1.8), 3
    rows = [em_coefficients(e, s) for e, s in zip(etas, sigs)]
    if len(set(rows)) == 1:
        rows = rows[:1]
    entry = "ebm_langevin_heun_chain_f32" if variant == "heun" else "ebm_langevin_chain_f32"
    seed = _rng.kernel_seed(1234 + n)
    n_kept = k // thin
    # native draws
    xa = x0.to(cuda_device).clone()
    ta_ = torch.full((n, n_kept, 2), float("nan"), device=cuda_device)
    _call(spec, xa, k, rows, clamp, thin, ta_, None, seed, entry)
    # the same field, injected
    noise = _field(n, seed, k, cuda_device)
    xb = x0.to(cuda_device).clone()
    tb = torch.full((n, n_kept, 2), float("nan"), device=cuda_device)
    _call(spec, xb, k, rows, clamp, thin, tb, noise, 0, entry)
    assert torch.equal(xa, xb) and torch.equal(ta_, tb)
    want_x, want_t, _ = oracle.langevin_chain(en, x0, noise.cpu(), etas, sigs, clamp=clamp, thin=thin, want_traj=True,
                                              integrator="heun" if variant == "heun" else "euler_maruyama")
    err = ((tb.cpu() - want_t).abs() / want_t.abs().clamp(min=1.0)).reshape(n, -1).amax(dim=1)
    assert (err <= 5e-5).float().mean().item() >= 0.95 and (err <= 5e-3).all(), err.max().item()
    assert ((xb.cpu() - want_x).abs() / want_x.abs().clamp(min=1.0)).max().item() <= 5e-3


def test_sampler_accepts_a_state_at_an_8_byte_offset(cuda_device):
    """The C ABI wants 16-byte aligned state pointers (EBM_EINVAL otherwise); a [n, 2] view that starts one row into
    its allocation is only 8-byte aligned -- the sampler hands the kernel an aligned copy and leaves the view alone."""
    model, _ = _models("gmm8", cuda_device)
    n, k = 1001, 6
    big = torch.randn(n + 1, 2, device=cuda_device)
    view = big[1:]
    assert view.data_ptr() % 16 == 8
    with pytest.raises(ValueError, match="16-byte aligned"):
        _call(model.fused_spec(), view, k, [em_coefficients(0.01, 1.0)], None, 1, None, None, 99)
    before = big.clone()
    s = ta.LangevinDynamics(model, step_size=0.01, device=cuda_device)
    a = s.sample(x=view, n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(4))
    b = s.sample(x=view.clone(), n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(4))
    assert torch.equal(a, b) and torch.equal(big, before)


@pytest.mark.parametrize("kind", ["gauss", "gmm8"])
@pytest.mark.parametrize("n", [3, 1500])
def test_pair_kernel_diagnostics_records(cuda_device, kind, n):
    model, _ = _models(kind, cuda_device)
    s = ta.LangevinDynamics(model, step_size=0.02, device=cuda_device)
    x0 = torch.randn(n, 2, device=cuda_device)
    gen = lambda: torch.Generator(device=cuda_device).manual_seed(3)
    traj = s.sample(x=x0, n_steps=9, thin=3, return_trajectory=True, generator=gen())
    out, diag = s.sample(x=x0, n_steps=9, thin=3, return_trajectory=True, return_diagnostics=True, generator=gen())
    assert torch.equal(out, traj)
    for j in range(3):
        xs = traj[:, j]
        assert torch.allclose(diag["mean"][j], xs.mean(0), atol=2e-6, rtol=1e-5)
        assert torch.allclose(diag["var"][j], xs.var(0, unbiased=False).clamp(1e-10, 1e10), atol=2e-6, rtol=2e-5)
        assert torch.allclose(diag["energy"][j], model(xs).mean(), atol=1e-5, rtol=1e-5)


def test_pair_layout_is_what_the_layout_query_reports(cuda_device):
    model, _ = _models("gauss", cuda_device)
    lay = _lib.diag_layout(model.fused_spec().to_c(), _lib.DIAG_LANGEVIN, 5000, 2, False, False)
    assert lay == ((5000 + 511) // 512, 2, 1024)
