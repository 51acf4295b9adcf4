#!/usr/bin/env python3
"""Shader-clock phases of the dense-Gaussian one-launch HMC kernel (dims 164 .. 256), debug build only:
    scripts/ab_build.sh PH gauss_hmc_stream.hip -DEBM_PHASE_TIMES && cp ab/PH.so torchebm_amd/libebm_hip.so   (then, on the GPU box)
    python scripts/hmc_stream_phase_times.py [dim]
Wave 0 of workgroup 0 adds up s_memtime differences per phase class over the first 20 passes (a pass = one piece of the force: two per
evaluation, three at eight tiles) in registers and writes the sums once -- no memory traffic in the loop; a stamp still drains the wave's LDS queue."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
from torchebm_amd import _lib
dim = int(sys.argv[1]) if len(sys.argv) > 1 else 256
nt = (dim + 31) // 32
dev = torch.device("cuda")
g = torch.Generator().manual_seed(dim)
a = torch.randn(dim, dim, generator=g)
model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
s = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=10, device=dev)
x = torch.randn(1 << 17, dim, device=dev)
for _ in range(3):
    s.sample(x=x, n_steps=1)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 16)()
assert _lib.lib().ebm_debug_hmc_phase_log(buf, 16) == 0
t = np.array(list(buf), dtype=np.int64)
passes = int(t[6])
names = ["entry: first operand (means, split, A reads)", "units in front of the sync", "sync (waitcnt + barrier)", "last unit of a stage (+ DMA requests, next operands)",
         "energy part", "between passes (kicks, drift, loop; the first: the prologue)"]
pieces = 3 if nt == 8 else 2
mf = 6 * 2 * nt * nt * 32 / pieces   # MFMA cycles per pass on average (an evaluation = 12 nt^2 MFMAs of 32 cycles, in `pieces` passes)
print(f"dim {dim}: ticks per PASS (average of {passes}; {pieces} passes = one evaluation); MFMA floor per pass {mf:.0f}")
tot = 0
for i, nm in enumerate(names):
    print(f"  {nm:62s} {t[i] / passes:9.0f}")
    tot += t[i] / passes
print(f"  {'sum':62s} {tot:9.0f}")
