"""HMC for dense Gaussians at widths that are not a multiple of 4, 21 .. 157, on SHIFTED rows (csrc/gauss_hmc_shift.hip,
mfma_hmc_body.h SH; the layout: tests/test_gauss_shift_gpu.py).  Injected momenta / uniforms against the oracle (accept
decisions identical wherever the oracle's margin is not borderline, states to the Gaussian tolerance) under every mass form,
with trajectory and records; the native draws equal to injected copies of the same field; the sampler takes it in one launch."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib
from torchebm_amd.integrators.symplectic import _mass_args

pytestmark = pytest.mark.gpu


def _model(dim, device, seed=0):
    g = torch.Generator().manual_seed(seed + dim)
    a = torch.randn(dim, dim, generator=g)
    mean = torch.randn(dim, generator=g) * 0.5
    cov = a @ a.t() / dim + 0.5 * torch.eye(dim)
    return ta.GaussianModel(mean, cov, device=device), oracle.Gaussian(mean, cov)


def _finish(layout, rec, n, dim, n_kept, dev):
    nb, S, E = layout
    out = {"mean": torch.empty(n_kept, dim, device=dev), "var": torch.empty(n_kept, dim, device=dev),
           "energy": torch.empty(n_kept, device=dev), "acceptance_rate": torch.empty(n_kept, device=dev)}
    work = torch.zeros(n_kept * (3 * dim + 3), dtype=torch.float64, device=dev)
    _lib.call("ebm_diag_finish_f32", rec.data_ptr(), n_kept, nb, S, E, n, dim, out["mean"].data_ptr(), out["var"].data_ptr(),
              out["energy"].data_ptr(), out["acceptance_rate"].data_ptr(), work.data_ptr(), _lib.stream_handle(dev))
    return {k: v.cpu() for k, v in out.items()}


@pytest.mark.parametrize("dim,n", [(21, 300), (30, 131), (33, 1), (50, 514), (51, 300), (70, 259), (99, 300), (126, 77), (130, 300), (157, 203),
                                   # shifted rows reaching 161 .. 256 tile coordinates: the streamed evaluation, one pre-split image per class
                                   (159, 150), (161, 300), (190, 131), (222, 97), (253, 203), (254, 300)])
@pytest.mark.parametrize("mass", [None, 2.5, "diag"])
def test_injected_draws_against_the_oracle(cuda_device, dim, n, mass):
    T, L, eps, thin = 4, 5, 0.06, 2
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
    o = oracle.hmc_chain(ref, x0, p, u, [eps] * T, L, mass=m, thin=thin, want_traj=True, want_diag=True, want_margins=True)
    m_dev = m.to(cuda_device) if torch.is_tensor(m) else m
    p_dev, u_dev = p.to(cuda_device), u.to(cuda_device)
    desc = model.fused_spec().to_c()
    K = 4 if dim % 2 else 2
    layout = _lib.diag_layout(desc, _lib.DIAG_HMC, n, dim, True, False)
    assert layout == (-(-n // (32 * K)) * K, dim, -32 * dim)
    states = []
    for with_records in (False, True):
        x = x0.to(cuda_device)
        mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
        traj = torch.full((n, T // thin, dim), float("nan"), device=cuda_device)
        kind, ms, md = _mass_args(m_dev, x)
        rec = torch.empty((T // thin) * layout[0] * (2 * dim + 8), device=cuda_device) if with_records else None
        _lib.call("ebm_hmc_chain_f32", desc, x.data_ptr(), n, dim, T, L, eps, None, kind, ms, _lib.ptr(md), thin, traj.data_ptr(),
                  _lib.ptr(rec), mask.data_ptr(), None, p_dev.data_ptr(), u_dev.data_ptr(), 0, 0, _lib.stream_handle(cuda_device))
        got_mask = mask.cpu().bool()
        safe = o["margins"].abs() > 1e-4
        assert torch.equal(got_mask[safe], o["accepted"][safe])
        rows_ok = (got_mask == o["accepted"]).all(dim=0)
        torch.testing.assert_close(x.cpu()[rows_ok], o["x"][rows_ok], rtol=5e-4, atol=5e-4)
        torch.testing.assert_close(traj.cpu()[rows_ok], o["trajectory"][rows_ok], rtol=5e-4, atol=5e-4)
        states.append(x.cpu())
        if with_records and bool(rows_ok.all()):
            d = _finish(layout, rec, n, dim, T // thin, cuda_device)
            want = o["diagnostics"]
            torch.testing.assert_close(d["acceptance_rate"], want["acceptance_rate"], rtol=0, atol=1e-6)
            torch.testing.assert_close(d["mean"], want["mean"], rtol=1e-3, atol=1e-3)
            torch.testing.assert_close(d["var"], want["var"], rtol=5e-3, atol=1e-4)
            torch.testing.assert_close(d["energy"], want["energy"], rtol=1e-3, atol=1e-3)
    assert torch.equal(states[0], states[1])  # records change nothing


@pytest.mark.parametrize("dim", [25, 50, 99, 150, 201, 254])
def test_native_draws_are_the_flat_field(cuda_device, dim):
    """Native RNG: momenta = the normal field at step offset + 2 t, uniforms = the uniform field at offset + 2 t + 1 --
    materialised with ebm_noise_fill_f32 and injected, the chains are bit-identical."""
    n, T, L, eps, seed, offset = 1003, 3, 4, 0.07, 91, 6
    model, _ = _model(dim, cuda_device, seed=3)
    st = _lib.stream_handle(cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device)
    p = torch.empty(T, n, dim, device=cuda_device)
    u = torch.empty(T, n, device=cuda_device)
    one = torch.empty(n * dim, device=cuda_device)
    one_u = torch.empty(n, device=cuda_device)
    for t in range(T):
        _lib.call("ebm_noise_fill_f32", one.data_ptr(), n * dim, 0, seed, offset + 2 * t, st)
        p[t] = one.view(n, dim)
        _lib.call("ebm_noise_fill_f32", one_u.data_ptr(), n, 1, seed, offset + 2 * t + 1, st)
        u[t] = one_u
    desc = model.fused_spec().to_c()
    out = []
    for inj in (False, True):
        x = x0.clone()
        mask = torch.empty(T, n, dtype=torch.uint8, device=cuda_device)
        _lib.call("ebm_hmc_chain_f32", desc, x.data_ptr(), n, dim, T, L, eps, None, 0, 1.0, None, 1, None, None, mask.data_ptr(), None,
                  p.data_ptr() if inj else None, u.data_ptr() if inj else None, seed, offset, st)
        out.append((x, mask))
    assert torch.equal(out[0][1], out[1][1]) and torch.equal(out[0][0], out[1][0])
    assert 0.3 < out[0][1].float().mean().item() <= 1.0


def test_the_sampler_takes_it_in_one_launch(cuda_device):
    model, _ = _model(150, cuda_device, seed=6)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=5, device=cuda_device)
    x0 = torch.randn(1 << 15, 150, device=cuda_device)
    c0 = hip_calls("ebm_hmc_chain_f32")
    out, diag = s.sample(x=x0, n_steps=6, thin=2, return_diagnostics=True)
    assert hip_calls("ebm_hmc_chain_f32") == c0 + 1
    assert torch.isfinite(out).all() and 0.5 < diag["acceptance_rate"].mean().item() <= 1.0
