#!/usr/bin/env python3
"""Training forward + gradient pass of the packaged MLPEnergy (dim 2 / 32, H 128) for kernel-trace A/B runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
dev = torch.device("cuda")
for dim, hidden in ((2, 128), (32, 128), (2, 64)):
    torch.manual_seed(0)
    m = ta.MLPEnergy(dim, hidden, device=dev)
    xx = torch.randn(131072, dim, device=dev)
    for _ in range(12):
        for p in m.parameters():
            p.grad = None
        m(xx).sum().backward()
torch.cuda.synchronize()
