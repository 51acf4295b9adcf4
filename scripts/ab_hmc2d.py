#!/usr/bin/env python3
"""HMC on the MLP energies, kernel ms per 10 transitions, L = 10, 65 536 chains (A/B helper: scripts/ab_hmc.sh)."""
import os, sys, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchebm_amd as ta
from torchebm_amd import _lib
dev = torch.device("cuda")
for dim, hidden in ((2, 128), (8, 128), (32, 128), (64, 128), (32, 64)):
    torch.manual_seed(0)
    m = ta.MLPEnergy(dim, hidden, device=dev)
    s = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=10, device=dev)
    x0 = torch.randn(65536, dim, device=dev)
    for _ in range(2):
        s.sample(x=x0, n_steps=10)
    _lib.timed_events["ebm_hmc_chain_f32"] = []
    for _ in range(5):
        s.sample(x=x0, n_steps=10)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in _lib.timed_events.pop("ebm_hmc_chain_f32"))
    print(json.dumps({"case": f"hmc mlp dim={dim} H={hidden} L=10 T=10", "kernel_ms": ts[len(ts) // 2]}))
