"""Native RNG: Philox4x32-10 known answers (CPU oracle) and, on the GPU, the kernel's field
against the oracle: raw bits and uniforms bit-exact, normals within a stated tolerance."""

import numpy as np
import pytest
import torch

import oracle
from torchebm_amd import _lib

# Random123 kat_vectors (philox4x32, 10 rounds): counter, key -> output
KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
    ((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
    ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
]


@pytest.mark.parametrize("ctr,key,want", KAT)
def test_philox_known_answers(ctr, key, want):
    got = oracle.philox4x32_10(*[[c] for c in ctr], key[0], key[1])
    assert tuple(int(g[0]) for g in got) == want


def test_oracle_normals_are_standard():
    z = oracle.normal_field(99, 3, 1 << 18).astype(np.float64)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01
    assert abs((z**3).mean()) < 0.03 and abs((z**4).mean() - 3.0) < 0.08


def _fill(n, kind, seed, step, device):
    out = torch.empty(n, dtype=torch.float32, device=device)
    _lib.call("ebm_noise_fill_f32", out.data_ptr(), n, kind, seed, step, _lib.stream_handle(device))
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("n", [1, 3, 4, 1023, 4096 + 2])
def test_kernel_field_matches_oracle(cuda_device, n):
    seed, step = 0x1234_5678_9ABC_DEF0, (7 << 32) + 11
    raw = _fill(n, _lib.NOISE_RAW_U32, seed, step, cuda_device).view(torch.int32).cpu().numpy().view(np.uint32)
    assert np.array_equal(raw, oracle.raw_field(seed, step, n))
    uni = _fill(n, _lib.NOISE_UNIFORM, seed, step, cuda_device).cpu().numpy()
    assert np.array_equal(uni, oracle.uniform_field(seed, step, n))
    assert uni.min() >= 0.0 and uni.max() < 1.0
    nor = _fill(n, _lib.NOISE_NORMAL, seed, step, cuda_device).cpu().numpy()
    want = oracle.normal_field(seed, step, n)
    # v_log/v_sqrt/v_sin/v_cos are ~1 ulp hardware approximations; v_sin/v_cos have an
    # absolute error of a few 1e-7 near their zeros, scaled by r <= 6.7
    np.testing.assert_allclose(nor, want, rtol=2e-5, atol=5e-6)


@pytest.mark.gpu
def test_kernel_normals_moments_and_independence(cuda_device):
    n = 1 << 22
    a = _fill(n, _lib.NOISE_NORMAL, 42, 0, cuda_device).double()
    b = _fill(n, _lib.NOISE_NORMAL, 42, 1, cuda_device).double()
    c = _fill(n, _lib.NOISE_NORMAL, 43, 0, cuda_device).double()
    for z in (a, b, c):
        assert abs(z.mean().item()) < 3e-3 and abs(z.std().item() - 1.0) < 3e-3
        assert abs((z**4).mean().item() - 3.0) < 0.03
    assert abs((a * b).mean().item()) < 3e-3 and abs((a * c).mean().item()) < 3e-3
    assert abs((a[:-1] * a[1:]).mean().item()) < 3e-3  # neighbours (same / adjacent counters)
    assert not torch.equal(a, b) and not torch.equal(a, c)
    assert torch.isfinite(a).all()
