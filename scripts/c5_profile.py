#!/usr/bin/env python3
"""Config-5 sampler call (step route: autograd gradient + fused HIP update) in a loop, for rocprofv3:
how much of the wall time is GPU-busy (sum of kernel durations) vs launch gaps."""
import os, sys, time
import torch
from torch import nn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchebm_amd as ta
from torchebm_amd.utils.synthetic import two_moons

dev = torch.device("cuda")
class MLPEnergy(ta.core.BaseModel):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(2, 128), nn.SiLU(), nn.Linear(128, 128), nn.SiLU(), nn.Linear(128, 1))
    def forward(self, x):
        return self.net(x).squeeze(-1)
torch.manual_seed(0)
model = MLPEnergy().to(dev)
data = two_moons(65536, 0.05, seed=0, device=dev)
s = ta.LangevinDynamics(model, step_size=0.1, device=dev)
for _ in range(3):
    s.sample(x=data, n_steps=20)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    s.sample(x=data, n_steps=20)
torch.cuda.synchronize()
print("wall per call ms:", (time.perf_counter() - t0) / 10 * 1e3)
