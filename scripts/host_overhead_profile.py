#!/usr/bin/env python3
"""cProfile of LangevinDynamics.sample() on a tiny batch (pure host overhead), for one energy."""
import cProfile, pstats, sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
dev = torch.device("cuda")
kind = sys.argv[1] if len(sys.argv) > 1 else "mlp"
model = ta.MLPEnergy(2, device=dev) if kind == "mlp" else ta.DoubleWellModel(device=dev)
s = ta.LangevinDynamics(model, step_size=0.01, device=dev)
x = torch.randn(1024, 2, device=dev)
for _ in range(20):
    s.sample(x=x, n_steps=10)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    s.sample(x=x, n_steps=10)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
