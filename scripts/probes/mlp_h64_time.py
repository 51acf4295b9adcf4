#!/usr/bin/env python3
"""Kernel time of the fused Langevin call on MLPEnergy(dim, 64) -- the H = 64 kernels (one or two workgroups per CU)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torchebm_amd as ta  # noqa: E402

dev = torch.device("cuda")
for dim, hidden in ((2, 64), (32, 64), (64, 64), (128, 64), (2, 128), (32, 128)):
    torch.manual_seed(0)
    m = ta.MLPEnergy(dim, hidden, device=dev)
    s = ta.LangevinDynamics(m, step_size=0.05, device=dev)
    x0 = torch.randn(131072, dim, device=dev)
    for _ in range(3):
        s.sample(x=x0, n_steps=20)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(10):
        s.sample(x=x0, n_steps=20)
    t1.record()
    torch.cuda.synchronize()
    print(f"dim {dim:4d} hidden {hidden:4d}: {t0.elapsed_time(t1) / 10:8.3f} ms per call (131072 chains, k = 20)")
