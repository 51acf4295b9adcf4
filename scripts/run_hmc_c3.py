#!/usr/bin/env python3
"""One shape of BASELINE config 3's kernel, a few launches (for rocprofv3 passes):
    python scripts/run_hmc_c3.py [ring|dense|plane4] [T] [L] [launches]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402

dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "ring"
T = int(sys.argv[2]) if len(sys.argv) > 2 else 50
L = int(sys.argv[3]) if len(sys.argv) > 3 else 20
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
n, dim = 1 << 18, 32
if which == "ring":
    model = ta.core.ring_mixture(8, dim, device=dev)
elif which == "plane4":
    m = torch.zeros(8, dim)
    m[:, :4] = torch.randn(8, 4, generator=torch.Generator().manual_seed(3)) * 2.0
    model = ta.GaussianMixtureModel(m, sigma=1.0, device=dev)
else:
    model = ta.GaussianMixtureModel(torch.randn(8, dim, generator=torch.Generator().manual_seed(7)) * 2.0, sigma=1.0, device=dev)
spec = model.fused_spec()
c = spec.to_c()
st = _lib.stream_handle(dev)
x = torch.randn(n, dim, device=dev).clamp_(-3, 3)
for _ in range(reps):
    _lib.call("ebm_hmc_chain_f32", c, x.data_ptr(), n, dim, T, L, 0.1, None, 0, 0.0, None, 1, None, None, None, None, None, None, 1, 0, st)
torch.cuda.synchronize()
print("ok", which, T, L, float(x.abs().mean()))
