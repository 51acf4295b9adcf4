"""Index width: launches whose flat element count reaches and passes 2^31 / 2^32, and BASELINE configs[3] WHOLE on one
GPU (2^23 x 128, k = 500: 2^30 elements, 4 GiB of state) -- its largest single-GPU form.  The rows at the far end of
the chain matrix must see the same (seed, step, element) field and the same arithmetic as the first ones:
  * the RNG field itself against the numpy Philox oracle in windows around element 2^31 and 2^32 (bit-exact);
  * the fused Langevin kernel at 2^31 + 2^16 elements, last rows bit-exact against the oracle fed with that field;
  * the per-step kernel beyond 2^31 elements;
  * whole config 4: the last 48 chains over all 500 steps, bit-exact (tests/test_config4_gpu.py does the first 48 of
    one shard)."""

import numpy as np
import pytest
import torch

import oracle
import torchebm_amd as ta
from helpers import hip_calls
from torchebm_amd import _lib, _rng
from torchebm_amd.samplers.langevin import em_coefficients

pytestmark = pytest.mark.gpu

ETA, SIGMA = 0.01, 1.0


def _need(cuda_device, gib):
    free, _ = torch.cuda.mem_get_info(cuda_device)
    if free < gib * (1 << 30):
        pytest.skip(f"needs {gib} GiB of free device memory, {free >> 30} GiB available")


def _raw_window(seed, step, first_elem, count):
    """uint32 field values of elements [first_elem, first_elem + count) from the numpy oracle (first_elem % 4 == 0)."""
    g = np.arange(first_elem // 4, (first_elem + count + 3) // 4, dtype=np.uint64)
    o = oracle.philox4x32_10(g & np.uint64(0xFFFFFFFF), g >> np.uint64(32), np.full(g.size, step & 0xFFFFFFFF, dtype=np.uint64),
                             np.full(g.size, (step >> 32) & 0xFFFFFFFF, dtype=np.uint64), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return np.stack(o, axis=1).reshape(-1)[:count]


def test_rng_field_past_2_31_and_2_32_elements(cuda_device):
    _need(cuda_device, 20)
    n = (1 << 32) + (1 << 16)
    seed, step = 0x0123_4567_89AB_CDEF, (3 << 32) + 5
    out = torch.empty(n, dtype=torch.float32, device=cuda_device)
    _lib.call("ebm_noise_fill_f32", out.data_ptr(), n, _lib.NOISE_RAW_U32, seed, step, _lib.stream_handle(cuda_device))
    for first in (0, (1 << 31) - 2048, (1 << 31), (1 << 32) - 2048, (1 << 32), n - 4096):
        got = out[first : first + 4096].view(torch.int32).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, _raw_window(seed, step, first, 4096)), first
    del out
    # the normals a chain kernel consumes there: same indexing path, values against the float64 Box-Muller of the oracle
    nor = torch.empty((1 << 31) + 4096, dtype=torch.float32, device=cuda_device)
    _lib.call("ebm_noise_fill_f32", nor.data_ptr(), nor.numel(), _lib.NOISE_NORMAL, seed, step, _lib.stream_handle(cuda_device))
    first = (1 << 31) - 2048
    raw = _raw_window(seed, step, first, 4096).reshape(-1, 4)
    want = np.empty((raw.shape[0], 4))
    for a, b in ((0, 1), (2, 3)):
        u1 = (raw[:, a].astype(np.float64) * 2.0**-32 + 2.0**-33).astype(np.float32).astype(np.float64)
        rev = (raw[:, b].astype(np.float32) * np.float32(2.0**-32)).astype(np.float64)
        r = np.sqrt(-2.0 * np.log(u1))
        want[:, a], want[:, b] = r * np.sin(2.0 * np.pi * rev), r * np.cos(2.0 * np.pi * rev)
    np.testing.assert_allclose(nor[first : first + 4096].cpu().numpy(), want.reshape(-1), rtol=2e-5, atol=5e-6)


def _field_rows(seed, k, n_elem, row0, rows, dim, device):
    """Normals of rows [row0, row0 + rows) for steps 0..k-1, cut out of the whole materialised field."""
    buf = torch.empty(n_elem, dtype=torch.float32, device=device)
    out = torch.empty(k, rows, dim)
    st = _lib.stream_handle(device)
    for i in range(k):
        _lib.call("ebm_noise_fill_f32", buf.data_ptr(), n_elem, _lib.NOISE_NORMAL, seed, i, st)
        out[i] = buf[row0 * dim : (row0 + rows) * dim].view(rows, dim).cpu()
    return out


def test_fused_langevin_beyond_2_31_elements(cuda_device):
    _need(cuda_device, 30)
    n, dim, k, rows = (1 << 23) + 256, 256, 5, 32          # 2^31 + 2^16 flat elements
    s = ta.LangevinDynamics(ta.DoubleWellModel(barrier_height=2.0, b=1.0, device=cuda_device), step_size=ETA, noise_scale=SIGMA,
                            device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(3)).clamp_(-3.0, 3.0)
    before = hip_calls("ebm_langevin_chain_f32")
    got = s.sample(x=x0, n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(41))
    assert hip_calls("ebm_langevin_chain_f32") == before + 1 and got.shape == (n, dim)
    seed = _rng.kernel_seed(41)
    for row0 in (0, (1 << 23) - 16, n - rows):     # the first rows, rows straddling element 2^31, the last rows
        noise = _field_rows(seed, k, n * dim, row0, rows, dim, cuda_device)
        want, _, _ = oracle.langevin_chain(oracle.DoubleWell(2.0, 1.0), x0[row0 : row0 + rows].cpu(), noise, [ETA] * k, [SIGMA] * k)
        assert torch.equal(got[row0 : row0 + rows].cpu(), want), row0
    assert torch.isfinite(got).all()


def test_step_kernel_beyond_2_31_elements(cuda_device):
    """ebm_langevin_step_f32 (the per-step route's update, config 5's kernel) over 2^31 + 2^12 elements with its own draws."""
    _need(cuda_device, 40)
    n_elem = (1 << 31) + 4096
    x = torch.randn(n_elem, device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(1))
    g = torch.randn(n_elem, device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(2))
    out = torch.empty_like(x)
    a, sq, coef = em_coefficients(ETA, SIGMA)
    st = _lib.stream_handle(cuda_device)
    _lib.call("ebm_langevin_step_f32", x.data_ptr(), g.data_ptr(), out.data_ptr(), None, n_elem, a, sq, coef, 0, 0.0, 0.0, 99, 7, st)
    noise = torch.empty(n_elem, device=cuda_device)
    _lib.call("ebm_noise_fill_f32", noise.data_ptr(), n_elem, _lib.NOISE_NORMAL, 99, 7, st)
    for first in (0, (1 << 31) - 2048, n_elem - 4096):
        sl = slice(first, first + 4096)
        xs, gs, ns = x[sl].cpu(), g[sl].cpu(), noise[sl].cpu()
        x1 = xs + torch.tensor(a) * (1.0 * (-gs))           # core/base_integrator.py:397, :728-729: separately rounded ops
        want = x1 + torch.tensor(coef) * (ns * torch.tensor(sq))
        assert torch.equal(out[sl].cpu(), want), first


def test_whole_config4_on_one_gpu(cuda_device):
    """BASELINE configs[3] unsharded: n_chains = 2^23, dim = 128, k = 500 in ONE launch on one GPU."""
    _need(cuda_device, 16)
    n, dim, k, rows = 1 << 23, 128, 500, 48
    s = ta.LangevinDynamics(ta.DoubleWellModel(barrier_height=2.0, b=1.0, device=cuda_device), step_size=ETA, noise_scale=SIGMA,
                            device=cuda_device)
    x0 = torch.randn(n, dim, device=cuda_device, generator=torch.Generator(device=cuda_device).manual_seed(1234)).clamp_(-4.0, 4.0)
    before = hip_calls("ebm_langevin_chain_f32")
    got = s.sample(x=x0, n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(77))
    assert hip_calls("ebm_langevin_chain_f32") == before + 1
    assert got.shape == (n, dim) and torch.isfinite(got).all()
    m = got.abs().mean().item()
    assert 0.80 < m < 0.92, m
    seed = _rng.kernel_seed(77)
    for row0 in (n - rows, n // 2 - 24):
        noise = _field_rows(seed, k, n * dim, row0, rows, dim, cuda_device)
        want, _, _ = oracle.langevin_chain(oracle.DoubleWell(2.0, 1.0), x0[row0 : row0 + rows].cpu(), noise, [ETA] * k, [SIGMA] * k)
        assert torch.equal(got[row0 : row0 + rows].cpu(), want), row0
    # a shard run on its own (rows of the second eighth) sees other draws only through the element index: the sharded
    # job seeds each rank differently (base + rank), so this is a statement about ONE launch -- rows [0, 2048) alone
    sub = s.sample(x=x0[:2048], n_steps=k, generator=torch.Generator(device=cuda_device).manual_seed(77))
    assert torch.equal(sub, got[:2048])
