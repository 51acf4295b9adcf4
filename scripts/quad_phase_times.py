#!/usr/bin/env python3
"""Per-phase shader clocks of the quad-wave MLP chain kernel (mlp_quad.hip built with -DEBM_PHASE_TIMES:
scripts/build_quad_ab.sh -> build/ab/quad_times.so / quad_solo.so, copied over the library): wave 0 of each chain tile of
workgroup 0 stamps the start of every phase, its end and the moment it leaves the tile barrier."""
import collections
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402

dev = torch.device("cuda")
dim = int(os.environ.get("Q_DIM", 2))
n, k = 65536, 20
torch.manual_seed(0)
model = ta.MLPEnergy(dim, 128, device=dev)
s = ta.LangevinDynamics(model, step_size=0.1, device=dev)
x0 = torch.randn(n, dim, device=dev)
for _ in range(3):
    s.sample(x=x0, n_steps=k)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
buf = (C.c_ulonglong * (2 * 512 * 3))()
assert lib.ebm_debug_quad_log(buf, len(buf)) == 0
t = torch.tensor(list(buf), dtype=torch.int64).view(2, 512, 3)
for g in range(2):
    if int(t[g, 6 * k - 1, 2]) == 0:
        continue
    tot = int(t[g, 6 * k - 1, 2] - t[g, 0, 0])
    print(f"tile {g}: whole loop {tot} cycles; {tot / k:.0f} per step")
    acc = collections.defaultdict(list)
    for tick in range(12, 6 * (k - 2)):
        acc[tick % 6].append((int(t[g, tick, 1] - t[g, tick, 0]), int(t[g, tick, 2] - t[g, tick, 1])))
    for ph in range(6):
        v = acc[ph]
        print(f"  phase {ph}: work {sum(a for a, _ in v) / len(v):7.0f}  barrier {sum(b for _, b in v) / len(v):7.0f}")
