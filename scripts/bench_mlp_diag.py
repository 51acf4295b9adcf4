#!/usr/bin/env python3
"""Cost of in-kernel diagnostics records on the MLP energy: config 5's sampler call (2-128-128-1,
65 536 chains, k = 20) and the benchmark network (dim 32), plain vs return_diagnostics=True at thin = 5 / 1; HMC L = 10."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchebm_amd as ta  # noqa: E402

dev = torch.device("cuda")


def timed(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n, k = 65536, 20
for dim in (2, 32):
    torch.manual_seed(0)
    m = ta.MLPEnergy(dim, 128, device=dev)
    s = ta.LangevinDynamics(m, step_size=0.1, device=dev)
    s.donate_input = False
    x0 = torch.randn(n, dim, device=dev)
    plain = timed(lambda: s.sample(x=x0, n_steps=k))
    row = {"case": f"langevin mlp {dim}-128-128-1 n={n} k={k}", "plain_ms": plain}
    for thin in (5, 1):
        t = timed(lambda: s.sample(x=x0, n_steps=k, thin=thin, return_diagnostics=True))
        row[f"diag_thin{thin}_ms"] = t
        row[f"diag_thin{thin}_over_plain"] = t / plain
    print(json.dumps(row), flush=True)
    h = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=10, device=dev)
    plain = timed(lambda: h.sample(x=x0, n_steps=10), reps=5, warm=2)
    t = timed(lambda: h.sample(x=x0, n_steps=10, thin=5, return_diagnostics=True), reps=5, warm=2)
    print(json.dumps({"case": f"hmc mlp {dim}-128-128-1 n={n} L=10 T=10", "plain_ms": plain, "diag_thin5_ms": t, "diag_thin5_over_plain": t / plain}),
          flush=True)
