#!/usr/bin/env python3
"""Overhead of the HMC per-transition route (everything except the model's gradient): a cheap energy on the step route,
n = 65 536 chains, L = 10, 10 transitions, over dims; per-kernel GPU time from the library's launch events."""
import os, sys, time, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
from torchebm_amd import _lib

dev = torch.device("cuda")


class Sub(ta.HarmonicModel):
    def forward(self, x):
        return super().forward(x)


def wall(fn, reps=5, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


n, T, L = 65536, 10, 10
for dim in (8, 32, 128):
    m = Sub(device=dev)
    x = torch.randn(n, dim, device=dev)
    row = {"dim": dim}
    for graph in (False, True):
        s = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=L, device=dev)
        s.capture_graph = graph
        row["graph_ms" if graph else "eager_ms"] = wall(lambda: s.sample(x=x, n_steps=T))
    s = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=L, device=dev)
    s.capture_graph = False
    names = [k for k in _lib._SIGS] if hasattr(_lib, "_SIGS") else []
    for k in ("ebm_leapfrog_kick_drift_f32", "ebm_leapfrog_kick_f32", "ebm_hmc_accept_f32", "ebm_noise_fill_f32"):
        _lib.timed_events[k] = []
    s.sample(x=x, n_steps=T)
    torch.cuda.synchronize()
    for k in ("ebm_leapfrog_kick_drift_f32", "ebm_leapfrog_kick_f32", "ebm_hmc_accept_f32", "ebm_noise_fill_f32"):
        ev = _lib.timed_events.pop(k)
        row[k] = {"calls": len(ev), "total_ms": sum(a.elapsed_time(b) for a, b in ev)}
    with torch.no_grad():
        g = wall(lambda: m.gradient(x), reps=20)
    row["gradient_ms"] = g
    row["state_MB"] = n * dim * 4 / 1e6
    print(json.dumps(row), flush=True)
