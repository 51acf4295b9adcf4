"""Dense Gaussians at widths that are not a multiple of 4, 17 .. 254, on SHIFTED rows (csrc/gauss_shift.hip, gauss_res_shift.hip,
gauss_mfma_body.h SH): a workgroup takes the chains of one alignment class c = K j + s and lays its tiles over the aligned
flat range that contains the row, so that a register quad is one float4 / one Philox counter of the flat field.  Against the
oracle on injected noise (every class, ragged chain counts, clamp, trajectory), the native field bit-identical to the
materialised one, records (interleaved classes: include/ebm_hip.h, negative block_elems) against the trajectory, neighbours
that hold NaN."""

import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients

pytestmark = pytest.mark.gpu

DIMS = [17, 19, 21, 25, 30, 33, 50, 51, 66, 70, 99, 126, 130, 157,
        159, 161, 190, 222, 253, 254]  # (from 159: the streamed kernel, one pre-split image per alignment class)


def _model(dim, device, seed=0):
    g = torch.Generator().manual_seed(seed + dim)
    a = torch.randn(dim, dim, generator=g)
    mean = torch.randn(dim, generator=g) * 0.5
    cov = a @ a.t() / dim + 0.5 * torch.eye(dim)
    return ta.GaussianModel(mean, cov, device=device), oracle.Gaussian(mean, cov)


def _chain(model, x, k, eta, noise=None, seed=0, offset=0, clamp=None, thin=1, traj=None, records=None):
    n, dim = x.shape
    a, sq, coef = em_coefficients(eta, 1.0)
    desc = model.fused_spec().to_c()
    _lib.call("ebm_langevin_chain_f32", desc, x.data_ptr(), n, dim, k, a, sq, coef, None, int(clamp is not None),
              clamp[0] if clamp else 0.0, clamp[1] if clamp else 0.0, thin, _lib.ptr(traj), _lib.ptr(records), _lib.ptr(noise), seed, offset,
              _lib.stream_handle(x.device))
    return x


@pytest.mark.parametrize("dim", DIMS)
@pytest.mark.parametrize("n", [1, 37, 515])
def test_injected_noise_against_the_oracle(cuda_device, dim, n):
    model, ref = _model(dim, cuda_device)
    g = torch.Generator().manual_seed(100 * dim + n)
    k = 6
    x0 = torch.randn(n, dim, generator=g)
    noise = torch.randn(k, n, dim, generator=g)
    want, _, _ = oracle.langevin_chain(ref, x0, noise, [0.02] * k, [1.0] * k)
    got = _chain(model, x0.to(cuda_device), k, 0.02, noise=noise.to(cuda_device)).cpu()
    torch.testing.assert_close(got, want, rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("dim", [21, 50, 99, 157, 161, 254])
def test_clamp_and_trajectory_against_the_oracle(cuda_device, dim):
    model, ref = _model(dim, cuda_device, seed=2)
    g = torch.Generator().manual_seed(dim)
    n, k, thin = 203, 8, 2
    x0 = torch.randn(n, dim, generator=g)
    noise = torch.randn(k, n, dim, generator=g)
    want, want_traj, _ = oracle.langevin_chain(ref, x0, noise, [0.05] * k, [1.0] * k, clamp=(-0.8, 0.9), thin=thin, want_traj=True)
    traj = torch.full((n, k // thin, dim), float("nan"), device=cuda_device)
    got = _chain(model, x0.to(cuda_device), k, 0.05, noise=noise.to(cuda_device), clamp=(-0.8, 0.9), thin=thin, traj=traj).cpu()
    torch.testing.assert_close(got, want, rtol=3e-5, atol=3e-5)
    torch.testing.assert_close(traj.cpu(), want_traj, rtol=3e-5, atol=3e-5)


@pytest.mark.parametrize("dim", DIMS)
def test_native_field_is_the_materialised_one(cuda_device, dim):
    """The FAST kernel (native Philox, normals behind the MFMAs) equals the plain kernel fed the field ebm_noise_fill_f32
    writes for the same (seed, step): the shifted tiles draw the flat field's counters."""
    model, _ = _model(dim, cuda_device, seed=3)
    n, k, seed, offset = 1003, 5, 77, 11
    x0 = torch.randn(n, dim, device=cuda_device)
    field = torch.empty(k, n, dim, device=cuda_device)
    one = torch.empty(n * dim, device=cuda_device)  # (a step of the [k, n, dim] field starts off the 16-byte grid)
    for s in range(k):
        _lib.call("ebm_noise_fill_f32", one.data_ptr(), n * dim, 0, seed, offset + s, _lib.stream_handle(cuda_device))
        field[s] = one.view(n, dim)
    a = _chain(model, x0.clone(), k, 0.02, seed=seed, offset=offset)
    b = _chain(model, x0.clone(), k, 0.02, noise=field)
    assert torch.equal(a, b)
    # the guard rows around the state stay untouched (partial quads at both ends of a row are element-wise stores)
    buf = torch.full((n + 2, dim), 7.0, device=cuda_device)
    pad = (-buf[1].data_ptr() // 4) % 4  # first 16-byte aligned element at or after row 1
    flat = buf.view(-1)
    x = flat[dim + pad: dim + pad + n * dim].view(n, dim)
    x.copy_(x0)
    _chain(model, x, k, 0.02, seed=seed, offset=offset)
    assert torch.equal(x, a)
    assert (flat[: dim + pad] == 7.0).all() and (flat[dim + pad + n * dim:] == 7.0).all()


@pytest.mark.parametrize("dim,n", [(30, 1000), (50, 514), (33, 129), (99, 1001), (126, 700), (150, 333), (161, 515), (222, 300), (253, 129)])
def test_records_interleave_the_classes_and_change_nothing(cuda_device, dim, n):
    model, _ = _model(dim, cuda_device, seed=4)
    layout = _lib.diag_layout(model.fused_spec().to_c(), _lib.DIAG_LANGEVIN, n, dim)
    K = 4 if dim % 2 else 2
    assert layout == (-(-n // (32 * K)) * K, dim, -32 * dim)
    s = ta.LangevinDynamics(model, step_size=0.02, noise_scale=0.8, device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device)
    c0 = hip_calls("ebm_langevin_chain_f32")
    traj, diag = s.sample(x=x0, n_steps=9, thin=3, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(3))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
    plain = s.sample(x=x0, n_steps=9, generator=torch.Generator(device=cuda_device).manual_seed(3))
    assert torch.equal(traj[:, -1], plain)
    t64 = traj.double()
    torch.testing.assert_close(diag["mean"].double(), t64.mean(dim=0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(diag["var"].double(), t64.var(dim=0, unbiased=False), rtol=1e-4, atol=1e-7)
    want_e = torch.stack([model(traj[:, j]).double().mean() for j in range(3)])
    torch.testing.assert_close(diag["energy"].double(), want_e, rtol=1e-4, atol=1e-5)


def test_a_nan_chain_does_not_reach_its_neighbours(cuda_device):
    """A tile's leading / trailing coordinates are the neighbouring chains' elements: loaded as 0, so that a non-finite
    neighbour cannot poison the contraction through 0 * NaN."""
    dim, n = 51, 130
    model, _ = _model(dim, cuda_device, seed=5)
    x0 = torch.randn(n, dim, device=cuda_device)
    clean = _chain(model, x0.clone(), 4, 0.02, seed=5)
    dirty0 = x0.clone()
    dirty0[64] = float("nan")
    dirty0[65, -1] = float("inf")
    dirty = _chain(model, dirty0, 4, 0.02, seed=5)
    keep = torch.ones(n, dtype=torch.bool, device=cuda_device)
    keep[64] = keep[65] = False
    assert torch.equal(dirty[keep], clean[keep])
    assert torch.isnan(dirty[64]).all() and not torch.isfinite(dirty[65]).all()


def test_the_sampler_keeps_the_chain_kernel_above_128(cuda_device):
    """Widths off multiples of 4 up to 254 have a chain kernel now: the sampler no longer reroutes them to the GEMM step route."""
    for dim in (150, 201):
        model, _ = _model(dim, cuda_device, seed=6)
        s = ta.LangevinDynamics(model, step_size=0.02, device=cuda_device)
        x0 = torch.randn(1 << 15, dim, device=cuda_device)
        c0 = hip_calls("ebm_langevin_chain_f32")
        s.sample(x=x0, n_steps=20)
        assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
