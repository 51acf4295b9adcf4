"""Oracle: the native RNG field of the kernels, restated in numpy.

Philox4x32-10 (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as 1, 2, 3",
SC'11; Random123 v1.14 ``philox.h``) -- a third-party algorithm the reference only reaches
through ``tl.randn`` in its Triton proof of concept (torchebm/cuda/fused_langevin.py:58,85).
Pinned by the Random123 known-answer vectors in tests/test_rng.py.

Field definition (include/ebm_hip.h): counter = (lo32(e/4), hi32(e/4), lo32(s), hi32(s)),
key = (lo32(seed), hi32(seed)); outputs o0..o3 feed Box-Muller pairs (o0,o1), (o2,o3).
"""

from __future__ import annotations

import numpy as np

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0: int, k1: int):
    """Vectorised over numpy uint32 arrays c0..c3; scalar key.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    k0, k1 = int(k0) & 0xFFFFFFFF, int(k1) & 0xFFFFFFFF
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        n0 = (p1 >> np.uint64(32)) ^ c1 ^ np.uint64(k0)
        n2 = (p0 >> np.uint64(32)) ^ c3 ^ np.uint64(k1)
        c1 = p1 & _MASK
        c3 = p0 & _MASK
        c0, c2 = n0, n2
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def raw_field(seed: int, step: int, n_elem: int) -> np.ndarray:
    """uint32[n_elem]: element e gets output (e % 4) of the counter for group e // 4."""
    n_groups = (n_elem + 3) // 4
    g = np.arange(n_groups, dtype=np.uint64)
    o = philox4x32_10(
        g & _MASK, g >> np.uint64(32),
        np.full(n_groups, step & 0xFFFFFFFF, dtype=np.uint64), np.full(n_groups, (step >> 32) & 0xFFFFFFFF, dtype=np.uint64),
        seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF,
    )
    return np.stack(o, axis=1).reshape(-1)[:n_elem]


def uniform_field(seed: int, step: int, n_elem: int) -> np.ndarray:
    """float32 in [0, 1): (o >> 8) * 2^-24."""
    return ((raw_field(seed, step, n_elem) >> np.uint32(8)).astype(np.float32) * np.float32(2.0**-24)).astype(np.float32)


def normal_field(seed: int, step: int, n_elem: int) -> np.ndarray:
    """Box-Muller in float64 on the same uniforms the kernel forms in fp32:
    u1 = fma(o_a, 2^-32, 2^-33) (rounded to fp32), angle = 2 pi * fp32(o_b * 2^-32).
    The kernel evaluates log/sqrt/sin/cos on the hardware transcendental unit, so agreement
    is to a few fp32 ulps of the result, not bit-exact (tolerance stated in the test)."""
    n_groups = (n_elem + 3) // 4
    raw = raw_field(seed, step, n_groups * 4).reshape(n_groups, 4)
    out = np.empty((n_groups, 4), dtype=np.float64)
    for a, b in ((0, 1), (2, 3)):
        u1 = (raw[:, a].astype(np.float64) * 2.0**-32 + 2.0**-33).astype(np.float32).astype(np.float64)
        rev = (raw[:, b].astype(np.float32) * np.float32(2.0**-32)).astype(np.float64)
        r = np.sqrt(-2.0 * np.log(u1))
        out[:, a] = r * np.sin(2.0 * np.pi * rev)
        out[:, b] = r * np.cos(2.0 * np.pi * rev)
    return out.reshape(-1)[:n_elem].astype(np.float32)
