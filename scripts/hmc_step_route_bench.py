#!/usr/bin/env python3
"""HMC on a neural energy (no fused kernel): eager step route vs one captured transition replayed."""
import json, os, sys, time
import torch
from torch import nn
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
from torchebm_amd.utils.synthetic import two_moons
dev = torch.device("cuda")


class Net(ta.BaseModel):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(2, 128), nn.SiLU(), nn.Linear(128, 128), nn.SiLU(), nn.Linear(128, 1))

    def forward(self, x):
        return self.net(x).squeeze(-1)


def wall(fn, reps=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


torch.manual_seed(0)
model = Net().to(dev)
n, T, L = 65536, 10, 10
x = two_moons(n, 0.05, seed=0, device=dev)
s = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=L, device=dev)
t_eager = wall(lambda: s.sample(x=x, n_steps=T))
s.capture_graph = True
t_graph = wall(lambda: s.sample(x=x, n_steps=T))
fused = ta.HamiltonianMonteCarlo(ta.MLPEnergy(2, device=dev), step_size=0.05, n_leapfrog_steps=L, device=dev)
t_fused = wall(lambda: fused.sample(x=x, n_steps=T))
flops = n * T * (L + 1) * 2 * (2 * 128 * 128 + 2 * 2 * 128)
print(json.dumps({"config": f"HMC on an MLP energy 2-128-128-1, n={n}, L={L}, {T} transitions per call",
                  "eager_step_route_s_per_call": t_eager, "graph_step_route_s_per_call": t_graph,
                  "fused_mlp_kernel_s_per_call": t_fused, "graph_speedup": t_eager / t_graph,
                  "fused_speedup_vs_eager": t_eager / t_fused, "mh_steps_per_s_fused": n * T / t_fused,
                  "fused_fp32_TFLOPs": flops / t_fused / 1e12}))
