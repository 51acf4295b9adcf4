#!/usr/bin/env python3
"""BASELINE config 3's kernel (HMC, 8-mode ring mixture, 2^18 x 32) over (T, L): separates the cost of a leapfrog
step from the per-transition work (momentum draw, exact energy, accept) and the per-launch work (prologue, the
pseudo-transition, the no-op launch of the other mixture kernel).  One JSON line per point; a least-squares fit
ms = a + T (b + c L) at the end.   python scripts/bench_hmc_c3_L.py [ring|dense]"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402

dev = torch.device("cuda")
which = sys.argv[1] if len(sys.argv) > 1 else "ring"
n, dim = 1 << 18, 32
if which == "ring":
    model = ta.core.ring_mixture(8, dim, device=dev)
else:
    model = ta.GaussianMixtureModel(torch.randn(8, dim, generator=torch.Generator().manual_seed(7)) * 2.0, sigma=1.0, device=dev)
c = model.fused_spec().to_c()
st = _lib.stream_handle(dev)
x = torch.randn(n, dim, device=dev).clamp_(-3, 3)


def time_call(T, L, reps=7):
    def run():
        _lib.call("ebm_hmc_chain_f32", c, x.data_ptr(), n, dim, T, L, 0.1, None, 0, 0.0, None, 1, None, None, None, None, None, None, 1, 0, st)
    run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        run()
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]


rows, ys = [], []
for T in (1, 10, 50):
    for L in (1, 5, 20, 40):
        ms = time_call(T, L)
        print(json.dumps({"mixture": which, "T": T, "L": L, "ms": round(ms, 4), "mh_steps_per_s": n * T / (ms * 1e-3)}), flush=True)
        rows.append([1.0, T, T * L])
        ys.append(ms)
coef, *_ = np.linalg.lstsq(np.array(rows), np.array(ys), rcond=None)
print(json.dumps({"mixture": which, "fit_ms": {"per_launch": round(float(coef[0]), 4), "per_transition": round(float(coef[1]), 5),
                                               "per_leapfrog_step": round(float(coef[2]), 5)}}))
