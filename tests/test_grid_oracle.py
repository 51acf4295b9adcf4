"""The SURVEY.md section 8c fixture grid (tests/golden/grid: every energy x {Langevin k=16, HMC L=5, HMC L=20} x dim in
{2, 32, 64, 100} at n = 1000, recorded from the reference by tests/golden/make_grid.py): the oracle reproduces the
reference's final state bit for bit (sha256 of the whole [1000, dim] tensor) and its diagnostics, from the seeds alone."""

import pytest
import torch

import oracle
from helpers import grid_inputs, grid_names, load_grid, oracle_energy, sha16

# the fixtures were recorded with one intra-op thread (tests/golden/make_golden.py): above 128 dims the CPU BLAS blocks the
# Gaussian's bmm differently per thread count, and the bit-for-bit bar below is against THAT run (7e-7 apart at 8 threads)
@pytest.fixture(autouse=True, scope="module")
def _one_intra_op_thread():
    """One thread for THIS module only; the previous count comes back on teardown (a module-level set_num_threads would
    make every later CPU test single-threaded and the outcome depend on collection order)."""
    before = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(before)


def test_the_grid_is_complete():
    names = set(grid_names())
    for dim in (2, 32, 64, 100):
        for tag in ("dw", "har", "gauss", "gmm8"):
            for kind in ("ld", "hmc5", "hmc20"):
                assert f"{kind}_{tag}_{dim}" in names
    assert {"ld_gmmd_32", "hmc5_gmmd_32", "hmc20_gmmd_32"} <= names
    assert {f"ld_gauss_{d}" for d in (5, 8, 12, 30, 50)} <= names   # round 3: the packed-row widths
    assert {"ld_gauss_160", "ld_gauss_256", "hmc5_gauss_30", "hmc5_gauss_256"} <= names and len(names) == 60   # round 3: above 128 dims


@pytest.mark.parametrize("name", grid_names("ld_"))
def test_oracle_langevin_grid(name):
    fx = load_grid(name)
    x0, noise = grid_inputs(fx)
    x, _, diag = oracle.langevin_chain(oracle_energy(fx["energy"]), x0, noise, fx["etas"], fx["sigmas"], thin=fx["thin"], want_diag=True)
    assert sha16(x) == fx["ref"]["sha_x"] and torch.equal(x[:256], fx["ref"]["x_rows"])
    for key in ("mean", "var", "energy"):
        assert torch.equal(diag[key], fx["ref"]["diagnostics"][key]), key


@pytest.mark.parametrize("name", grid_names("hmc"))
def test_oracle_hmc_grid(name):
    fx = load_grid(name)
    x0, p, u = grid_inputs(fx)
    o = oracle.hmc_chain(oracle_energy(fx["energy"]), x0, p, u, fx["eps"], fx["L"], mass=fx["mass"], thin=fx["thin"], want_diag=True)
    assert sha16(o["x"]) == fx["ref"]["sha_x"] and torch.equal(o["accepted"], fx["accepted"])
    for key in ("mean", "var", "energy", "acceptance_rate"):
        assert torch.equal(o["diagnostics"][key], fx["ref"]["diagnostics"][key]), key
