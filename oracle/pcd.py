"""Oracle (TEST INFRASTRUCTURE -- only tests/, smoke() and bench.py's cpu_baseline leg may import this): the chain starts of a
persistent-CD step, restated in numpy.

Reference: torchebm/core/base_loss.py:305-332 -- stratified reads of the replay buffer (row ``i stride + randint(stride)``) and
exploration noise ``+ 0.01 randn`` on the rows ``randperm(batch)[:n_new]``.  The reference draws through torch's generator; the
kernels (``ebm_pcd_start_points_f32``) draw from their own Philox field (oracle/philox.py), and choose the random subset without a
sort: { i : pi(i) < n_noise } for a keyed bijection pi of [0, batch) -- a six-round Feistel network on ceil(log2 batch) bits with
cycle walking.  This file restates THAT construction bit for bit (offsets and subset are integer work: exact; the normals are
Box-Muller on the hardware transcendental unit: a few fp32 ulps, tolerance in the test); that the construction has the reference's
LAW -- a uniformly random subset of exactly n_new rows -- is what the statistical tests check.
"""

from __future__ import annotations

import numpy as np

from . import philox


def _mix32(h: np.ndarray) -> np.ndarray:
    h = h.astype(np.uint32)
    h ^= h >> np.uint32(16)
    h = (h.astype(np.uint64) * np.uint64(0x21F0AAAD)).astype(np.uint32)
    h ^= h >> np.uint32(15)
    h = (h.astype(np.uint64) * np.uint64(0x735A2D97)).astype(np.uint32)
    h ^= h >> np.uint32(15)
    return h


def _feistel6(v: np.ndarray, a_bits: int, b_bits: int, keys) -> np.ndarray:
    left = (v >> np.uint32(b_bits)).astype(np.uint32)
    right = (v & np.uint32((1 << b_bits) - 1)).astype(np.uint32)
    for r in range(6):
        if r % 2 == 0:
            left ^= _mix32(right ^ np.uint32(keys[r])) >> np.uint32(32 - a_bits)
        else:
            right ^= _mix32(left ^ np.uint32(keys[r])) >> np.uint32(32 - b_bits)
    return (left << np.uint32(b_bits)) | right


def permutation(seed: int, step: int, batch: int) -> np.ndarray:
    """pi(0..batch-1): the keyed bijection of [0, batch) the kernel walks (round keys: the first six words of step + 1)."""
    bits = 2
    while bits < 31 and (1 << bits) < batch:
        bits += 1
    a_bits = bits // 2
    b_bits = bits - a_bits
    keys = philox.raw_field(seed, step + 1, 8)[:6]
    p = np.arange(batch, dtype=np.uint32)
    todo = np.ones(batch, dtype=bool)
    while todo.any():
        p[todo] = _feistel6(p[todo], a_bits, b_bits, keys)
        todo &= p >= batch
    return p


def start_points(buffer: np.ndarray, batch: int, stride: int, n_noise: int, noise_scale: float, seed: int, step: int):
    """``(out[batch, dim], rows[batch], noisy[batch])`` of ebm_pcd_start_points_f32 on a ``[buffer_size, dim]`` float32 buffer."""
    buffer = np.asarray(buffer, dtype=np.float32)
    size, dim = buffer.shape
    o = philox.raw_field(seed, step, batch).astype(np.uint64)
    r = (o * np.uint64(stride)) >> np.uint64(32)
    rows = ((np.arange(batch, dtype=np.uint64) * np.uint64(stride) + r) % np.uint64(size)).astype(np.int64)
    noisy = permutation(seed, step, batch) < n_noise
    z = philox.normal_field(seed, step + 2, batch * dim).reshape(batch, dim)
    out = buffer[rows].copy()
    bump = (z * np.float32(noise_scale)).astype(np.float32)
    out[noisy] = (out[noisy] + bump[noisy]).astype(np.float32)
    return out, rows, noisy
