#!/usr/bin/env python3
"""CPU baselines for configs 3 and 5: the oracle's restatement of the reference's CPU path (what the
reference itself executes: torch CPU ops + autograd), timed on this box's host cores on a bounded
sample of each workload.  One JSON line per config.  (Config 2's baseline is in bench.py.)"""
import json
import os
import sys
import time

import torch
from torch import nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402
import torchebm_amd as ta  # noqa: E402


def best_threads(fn):
    ncpu = os.cpu_count() or 1
    best, best_t = 1, float("inf")
    # EBM_CPU_THREADS="16,64" restricts the probe (on a 256-core GPU host the full sweep takes minutes)
    probe = [int(t) for t in os.environ.get("EBM_CPU_THREADS", "4,8,16,32,64,%d" % ncpu).split(",")]
    for th in sorted({t for t in probe if t <= ncpu}):
        torch.set_num_threads(th)
        fn()
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        if dt < best_t:
            best, best_t = th, dt
        if dt > 8:
            break
    torch.set_num_threads(best)
    return best, best_t


# ---- config 3: HMC L=20 on the 8-mode mixture, dim=32; sample: n = 2^13 chains, 2 transitions
n, dim, L, T = 1 << 13, 32, 20, 2
g = torch.Generator().manual_seed(0)
means = ta.core.ring_mixture(8, dim).means
en = oracle.GaussianMixture(means, 1.0)
x0 = torch.randn(n, dim, generator=g)


def hmc_run():
    p = torch.randn(T, n, dim, generator=g)
    u = torch.rand(T, n, generator=g)
    return oracle.hmc_chain(en, x0, p, u, [0.1] * T, L)


th, dt = best_threads(hmc_run)
times = []
for _ in range(3):
    t0 = time.perf_counter()
    hmc_run()
    times.append(time.perf_counter() - t0)
t = sorted(times)[1]
print(json.dumps({"config": "c3 cpu baseline (oracle: reference HMC loop, autograd gradient)", "threads": th, "cores": os.cpu_count(),
                  "sample": f"n=2^13 dim=32 L=20 T={T}", "s_per_run": t, "mh_steps_per_s": n * T / t,
                  "grad_evals_per_s": n * T * (2 * L) / t}), flush=True)

# ---- config 5: one PCD sampler call, MLP 2-128-128-1, n = 65536, k = 20 (full size)
torch.manual_seed(0)


class MLPEnergy(ta.core.BaseModel):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(2, 128), nn.SiLU(), nn.Linear(128, 128), nn.SiLU(), nn.Linear(128, 1))

    def forward(self, x):
        return self.net(x).squeeze(-1)


model = MLPEnergy()
n5, k5 = 65536, 20
x5 = torch.randn(n5, 2, generator=g)


def cd_run():
    x = x5
    for _ in range(k5):
        eps = torch.randn(n5, 2, generator=g)
        x = oracle.em_step(x, model.gradient(x), eps, 0.1, 1.0)
    return x


th, dt = best_threads(cd_run)
times = []
for _ in range(3):
    t0 = time.perf_counter()
    cd_run()
    times.append(time.perf_counter() - t0)
t = sorted(times)[1]
print(json.dumps({"config": "c5 cpu baseline (oracle: reference Langevin loop on the MLP energy, autograd gradient)", "threads": th,
                  "cores": os.cpu_count(), "sample": f"n=65536 dim=2 k={k5}", "s_per_sampler_call": t,
                  "chain_steps_per_s": n5 * k5 / t}), flush=True)
