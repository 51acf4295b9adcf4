#!/usr/bin/env python3
"""Every fused-MLP kernel family in one process, for scripts/ab_kt_all.sh (rocprofv3 --kernel-trace --stats around it): Langevin chains
over input widths / hidden sizes, HMC transitions, the training forward + gradient pass, the plain evaluation."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta

dev = torch.device("cuda")
n = 65536
for dim, hidden in ((2, 128), (8, 128), (32, 128), (64, 128), (100, 128), (128, 128), (2, 64), (32, 64), (64, 64)):
    torch.manual_seed(0)
    m = ta.MLPEnergy(dim, hidden, device=dev)
    x = torch.randn(n, dim, device=dev)
    s = ta.LangevinDynamics(m, step_size=0.05, device=dev)
    for _ in range(6):
        s.sample(x=x, n_steps=20)
    if dim in (2, 32):
        for _ in range(4):
            s.sample(x=x, n_steps=20, return_diagnostics=True)
    if dim in (8, 32, 64) or (dim, hidden) == (2, 128):
        h = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=10, device=dev)
        for _ in range(4):
            h.sample(x=x, n_steps=5)
    if dim in (2, 32):
        xx = torch.randn(2 * n, dim, device=dev)
        for _ in range(6):
            for p in m.parameters():
                p.grad = None
            m(xx).sum().backward()
torch.cuda.synchronize()
