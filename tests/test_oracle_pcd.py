"""The numpy restatement of the PCD start-point construction (oracle/pcd.py) on its own: the keyed Feistel walk is a bijection of
[0, batch) for every batch, the subset it selects has exactly n rows, and over keys every row is selected equally often."""

import numpy as np
import pytest

from oracle import pcd, philox


@pytest.mark.parametrize("batch", [1, 2, 3, 4, 5, 17, 259, 1000, 1024, 1025, 65536])
def test_permutation_is_a_bijection(batch):
    for step in (0, 5):
        p = pcd.permutation(0xABCDEF, step, batch)
        assert p.dtype == np.uint32 and np.array_equal(np.sort(p), np.arange(batch, dtype=np.uint32))
    assert batch < 4 or not np.array_equal(pcd.permutation(1, 0, batch), pcd.permutation(1, 3, batch))


def test_selected_subset_is_exact_and_uniform_over_keys():
    import scipy.stats as st

    batch, n, steps = 300, 30, 600
    hits = np.zeros(batch)
    for s in range(steps):
        sel = pcd.permutation(7, 3 * s, batch) < n
        assert sel.sum() == n
        hits += sel
    p = n / batch
    chi2 = ((hits - steps * p) ** 2 / (steps * p * (1 - p))).sum()
    assert 1e-4 < st.chi2.cdf(chi2, batch - 1) < 1 - 1e-4, chi2


def test_start_points_rows_are_stratified_and_noise_lands_on_the_subset():
    rng = np.random.default_rng(0)
    buf = rng.standard_normal((1000, 3)).astype(np.float32)
    out, rows, noisy = pcd.start_points(buf, 250, 4, 12, 0.01, seed=5, step=9)
    assert noisy.sum() == 12 and ((rows // 4) == np.arange(250)).all()
    assert np.array_equal(out[~noisy], buf[rows][~noisy])
    d = (out[noisy] - buf[rows][noisy]) / 0.01
    z = philox.normal_field(5, 11, 750).reshape(250, 3)[noisy]
    np.testing.assert_allclose(d, z, atol=2e-3)  # (the sum rounds to fp32 at |buf| ~ 1: 6e-8 / 0.01)
