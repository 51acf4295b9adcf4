"""Dense Gaussians at widths below 17 (from 17 on, widths off multiples of 4 run on SHIFTED rows: tests/test_gauss_shift_gpu.py): `pack`
consecutive chains of the row-major state run as ONE row of the block-diagonal Gaussian kron(I, Ps) (csrc/gauss_mfma.hip,
gauss_pack_factor) -- same flat element order, hence the same Philox field and the same update of every element.  The
reference's recorded runs at dims 5 / 8 / 12 / 30 / 50 are in tests/test_grid_gpu.py (ld_gauss_*); here: the sampler's
routes (trajectory scatter, pooled diagnostics) and the agreement with the lane-group kernel on the shared field."""

import pytest
import torch

import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib

pytestmark = pytest.mark.gpu


def _model(dim, device, seed=0):
    g = torch.Generator().manual_seed(seed + dim)
    a = torch.randn(dim, dim, generator=g)
    return ta.GaussianModel(torch.randn(dim, generator=g) * 0.5, a @ a.t() / dim + 0.5 * torch.eye(dim), device=device)


@pytest.mark.parametrize("dim,n,pack", [(8, 4096, 4), (16, 1000, 2), (5, 1024, 4), (14, 1000, 2), (10, 514, 2), (3, 4096, 8), (12, 1002, 2)])
def test_layout_query_says_packed_and_the_sampler_pools_the_records(cuda_device, dim, n, pack):
    model = _model(dim, cuda_device)
    layout = _lib.diag_layout(model.fused_spec().to_c(), _lib.DIAG_LANGEVIN, n, dim)
    assert layout == ((n // pack + 31) // 32, pack * dim, 32 * pack * dim)
    s = ta.LangevinDynamics(model, step_size=0.02, noise_scale=0.8, device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device)
    c0 = hip_calls("ebm_langevin_chain_f32")
    traj, diag = s.sample(x=x0, n_steps=9, thin=3, return_trajectory=True, return_diagnostics=True,
                          generator=torch.Generator(device=cuda_device).manual_seed(3))
    assert hip_calls("ebm_langevin_chain_f32") == c0 + 1
    plain = s.sample(x=x0, n_steps=9, generator=torch.Generator(device=cuda_device).manual_seed(3))
    assert torch.equal(traj[:, -1], plain)                       # trajectory rows land on their chains; records change nothing
    t64 = traj.double()
    torch.testing.assert_close(diag["mean"].double(), t64.mean(dim=0), rtol=1e-5, atol=1e-6)
    torch.testing.assert_close(diag["var"].double(), t64.var(dim=0, unbiased=False), rtol=1e-4, atol=1e-7)
    want_e = torch.stack([model(traj[:, j]).double().mean() for j in range(3)])
    torch.testing.assert_close(diag["energy"].double(), want_e, rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("dim", [8, 14])
def test_packed_rows_and_the_lane_group_kernel_share_the_field(cuda_device, dim):
    """n divisible by the pack factor: packed rows on the matrix cores; one chain more: the lane-group kernel.  Same seed ->
    same (seed, step, element) field -> the common chains agree to fp32 round-off of the two contractions."""
    model = _model(dim, cuda_device, seed=1)
    s = ta.LangevinDynamics(model, step_size=0.02, device=cuda_device)
    x0 = torch.randn(2049, dim, device=cuda_device)
    a = s.sample(x=x0[:2048], n_steps=20, generator=torch.Generator(device=cuda_device).manual_seed(5))
    b = s.sample(x=x0, n_steps=20, generator=torch.Generator(device=cuda_device).manual_seed(5))
    torch.testing.assert_close(a, b[:2048], rtol=2e-5, atol=2e-5)


def test_a_population_far_from_the_origin_keeps_its_variance(cuda_device):
    """The pooling is within-group variance + variance of the group means: conditioned by the spread, not by |mean|."""
    dim, n = 8, 4096
    model = ta.GaussianModel(torch.full((dim,), 1000.0), torch.eye(dim) * 0.0025, device=cuda_device)
    s = ta.LangevinDynamics(model, step_size=1e-4, noise_scale=1e-3, device=cuda_device)
    x0 = 1000.0 + 0.05 * torch.randn(n, dim, device=cuda_device)
    out, diag = s.sample(x=x0, n_steps=2, return_diagnostics=True)
    torch.testing.assert_close(diag["var"][-1].double(), out.double().var(dim=0, unbiased=False), rtol=2e-3, atol=0)
