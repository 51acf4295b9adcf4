#!/usr/bin/env python3
"""Config 5's training step as one HIP graph (utils.GraphedTrainingStep): 40 replays, for rocprofv3 --kernel-trace --stats
(which kernels the 2 ms of a step are).  C5_RATIO sets the loss's new_sample_ratio (default: the loss's own 0.05)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchebm_amd as ta  # noqa: E402
from torchebm_amd.utils import GraphedTrainingStep  # noqa: E402
from torchebm_amd.utils.synthetic import two_moons  # noqa: E402

dev = torch.device("cuda")
n, k = 65536, 20
data = two_moons(n, 0.05, seed=0, device=dev)
torch.manual_seed(0)
model = ta.MLPEnergy(2, device=dev)
sampler = ta.LangevinDynamics(model, step_size=0.1, noise_scale=1.0, device=dev)
kw = {}
if "C5_RATIO" in os.environ:
    kw["new_sample_ratio"] = float(os.environ["C5_RATIO"])
pcd = ta.ContrastiveDivergence(model, sampler, k_steps=k, persistent=True, buffer_size=n, init_steps=0, device=dev, **kw)
step = GraphedTrainingStep(pcd, torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True, **({"fused": True} if os.environ.get("C5_FUSED_ADAM") else {})),
                           enabled=os.environ.get("C5_EAGER") is None)
for _ in range(5):
    step(data)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(40):
    step(data)
torch.cuda.synchronize()
print("ms per training step:", (time.perf_counter() - t0) / 40 * 1e3, "replays", step.replays)
