#!/usr/bin/env python3
"""Per-phase shader-clock times of the wide-MLP Langevin kernel (debug build only):
    cd torchebm_amd/csrc && touch mlp_wide.hip && make CXXFLAGS_EXTRA=-DEBM_PHASE_TIMES
    python scripts/mlp_phase_times.py [dim] [hidden]          (on the GPU box)
Wave 0 of workgroup 0 stamps s_memtime at: loop top | state split | W1 x | W2 h1 | W2^T d2 | W1^T d1 | (loop top: update)."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402

dim = int(sys.argv[1]) if len(sys.argv) > 1 else 32
hidden = int(sys.argv[2]) if len(sys.argv) > 2 else 128
dev = torch.device("cuda")
n, k = 65536, 20
torch.manual_seed(0)
s = ta.LangevinDynamics(ta.MLPEnergy(dim, hidden, device=dev), step_size=0.05, device=dev)
x0 = torch.randn(n, dim, device=dev)
for _ in range(3):
    s.sample(x=x0, n_steps=k)
torch.cuda.synchronize()
NS = 6
buf = (ctypes.c_ulonglong * (NS * k))()
rc = _lib.lib().ebm_debug_phase_log(buf, NS * k)
assert rc == 0, rc
t = np.array(list(buf), dtype=np.int64).reshape(k, NS)
names = ["state split", "W1 x (+E1 tile 0)", "W2 h1 (+E1, E2 tile 0)", "W2^T d2 (+E2, E3 tile 0)", "W1^T d1 (+E3)", "update + noise"]
d = np.diff(t, axis=1)
upd = t[1:, 0] - t[:-1, NS - 1]
print(f"dim {dim} hidden {hidden}: median ticks per phase over {k} steps (one wave per SIMD)")
tot = 0
for i in range(NS - 1):
    m = float(np.median(d[:, i]))
    tot += m
    print(f"  {names[i]:28s} {m:9.0f}")
m = float(np.median(upd))
tot += m
print(f"  {names[-1]:28s} {m:9.0f}")
print(f"  {'sum':28s} {tot:9.0f}")
