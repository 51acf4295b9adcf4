#!/usr/bin/env python3
"""fp64 referee for EVERY HMC grid fixture (tests/golden/grid/hmc*.pt; round 4 had the quartic well only).

The HMC state is a tolerance tier: 160 leapfrog steps in the double well amplify a last-bit difference of the force
sum by up to 1e4, so "the kernel differs from the reference's fp32 run by 5e-4" says little about which of the two is
closer to the dynamics.  This script runs the SAME transitions in float64 -- the oracle (pinned bit for bit to the
reference's fp32 run when the fixtures were recorded: make_grid.py asserts torch.equal) on the same replayed draws,
upcast, following the recorded accept decisions -- and stores the first 256 rows next to the grid.  The GPU test then
requires the kernel to be as close to the fp64 chain as the reference's own fp32 arithmetic is (x4).

    python tests/golden/make_referee.py
"""
import glob
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

import oracle  # noqa: E402
from helpers import grid_inputs, oracle_energy, to64  # noqa: E402

OUT = os.path.join(HERE, "grid_referee")
os.makedirs(OUT, exist_ok=True)
torch.set_num_threads(1)

for path in sorted(glob.glob(os.path.join(HERE, "grid", "hmc*.pt"))):
    fx = torch.load(path, weights_only=False)
    x0, p, u = grid_inputs(fx)
    mass = fx["mass"]
    m64 = mass.double() if torch.is_tensor(mass) else mass
    f32 = oracle.hmc_chain(oracle_energy(fx["energy"]), x0, p, u, fx["eps"], fx["L"], mass=mass)
    assert torch.equal(f32["x"][:256], fx["ref"]["x_rows"]) and torch.equal(f32["accepted"], fx["accepted"]), fx["name"]
    f64 = oracle.hmc_chain(to64(oracle_energy(fx["energy"])), x0.double(), p.double(), u.double(), fx["eps"], fx["L"], mass=m64,
                           forced_accept=fx["accepted"])
    err = (f32["x"][:256].double() - f64["x"][:256]).abs().amax(dim=1)
    torch.save({"name": fx["name"], "x_rows_f64": f64["x"][:256].clone(), "ref_f32_err_median": float(err.median()),
                "ref_f32_err_max": float(err.max())}, os.path.join(OUT, fx["name"] + ".pt"))
    print(f"{fx['name']:16s} |ref_f32 - f64| per chain: median {err.median():.2e}  max {err.max():.2e}")
