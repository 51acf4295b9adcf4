#!/usr/bin/env python3
"""Fused MLP-energy Langevin chain over the reference's benchmark input widths (benchmarks/registry.py:372-387:
Linear(dim, 128) - SiLU - Linear(128, 128) - SiLU - Linear(128, 1) at dim 8 / 32 / 128): kernel ms per 20-step call,
exact-f32 matrix TFLOP/s and the fraction of the 157.3 TFLOP/s MFMA peak, next to the autograd step route."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402

dev = torch.device("cuda")
PEAK = 157.3
n, k = 65536, int(os.environ.get("MLP_K", "20"))  # MLP_K=200: the per-step cost without the launch + staging share
CASES = ((2, 128), (8, 128), (32, 128), (128, 128), (32, 64), (8, 256), (32, 256), (128, 256))
if os.environ.get("MLP_CASES"):  # e.g. MLP_CASES=32x256,128x256
    CASES = tuple(tuple(int(v) for v in c.split("x")) for c in os.environ["MLP_CASES"].split(","))
for dim, hidden in CASES:
    torch.manual_seed(0)
    model = ta.MLPEnergy(dim, hidden, device=dev)
    s = ta.LangevinDynamics(model, step_size=0.05, device=dev)
    x0 = torch.randn(n, dim, device=dev)
    for _ in range(2):
        s.sample(x=x0, n_steps=k)
    _lib.timed_events["ebm_langevin_chain_f32"] = []
    for _ in range(5):
        s.sample(x=x0, n_steps=k)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in _lib.timed_events.pop("ebm_langevin_chain_f32"))
    ms = ts[len(ts) // 2]
    dpad = 32 * ((dim + 31) // 32)
    useful = n * k * 2 * (2 * hidden * hidden + 2 * dim * hidden)          # the four contractions, real widths
    issued = n * k * 2 * (2 * hidden * hidden + 2 * dpad * hidden) if dpad > 4 else useful
    if os.environ.get("MLP_NO_STEP_ROUTE"):  # A/B runs (scripts/ab_mlp.sh): the fused kernel only
        print(json.dumps({"case": f"mlp_langevin dim={dim} H={hidden} n={n} k={k}", "kernel_ms": ms}), flush=True)
        continue
    # the autograd step route on the same network (HIP-graph replay, the default)
    class Sub(ta.MLPEnergy):
        def forward(self, x):
            return super().forward(x)
    sm = Sub(dim, hidden, device=dev)
    sm.load_state_dict(model.state_dict())
    ss = ta.LangevinDynamics(sm, step_size=0.05, device=dev)
    for _ in range(2):
        ss.sample(x=x0, n_steps=k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        ss.sample(x=x0, n_steps=k)
    torch.cuda.synchronize()
    t_step = (time.perf_counter() - t0) / 3 * 1e3
    print(json.dumps({"case": f"mlp_langevin dim={dim} H={hidden} n={n} k={k}", "kernel_ms": ms,
                      "useful_TFLOPs": useful / ms / 1e9, "frac_of_f32_mfma_peak": useful / ms / 1e9 / PEAK,
                      "issued_TFLOPs_incl_padding": issued / ms / 1e9, "autograd_step_route_ms": t_step,
                      "speedup_vs_step_route": t_step / ms}), flush=True)
