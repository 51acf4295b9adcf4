#!/usr/bin/env python3
"""A/B of the quad-wave MLP chain kernel (mlp_quad.hip) against mlp_wide_chain_kernel<4, 1, 2, 1> in ONE process:
needs a library whose mlp_wide.o was built with -DEBM_AB_SWITCHES (build/ab/quad.so copied over
torchebm_amd/libebm_hip.so); EBM_MLP_NO_QUAD=1 routes the plain call to the round-3 kernel.
Prints per input width: kernel ms of both, and the largest difference of the final states (same draws)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402

dev = torch.device("cuda")
n, k = int(os.environ.get("AB_N", 65536)), int(os.environ.get("AB_K", 20))


def run(model, x0, quad, reps=7, **kw):
    os.environ["EBM_MLP_NO_QUAD"] = "0" if quad else "1"
    s = ta.LangevinDynamics(model, step_size=0.1, device=dev)
    out = None
    for _ in range(2):
        out = s.sample(x=x0, n_steps=k, generator=torch.Generator(device=dev).manual_seed(7), **kw)
    _lib.timed_events["ebm_langevin_chain_f32"] = []
    for _ in range(reps):
        s.sample(x=x0, n_steps=k, generator=torch.Generator(device=dev).manual_seed(7), **kw)
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in _lib.timed_events.pop("ebm_langevin_chain_f32"))
    return out, ts[len(ts) // 2], ts[0]


for dim in [int(v) for v in os.environ.get("AB_DIMS", "2,4,8,16,20,32").split(",")]:
    torch.manual_seed(0)
    model = ta.MLPEnergy(dim, 128, device=dev)
    x0 = torch.randn(n, dim, device=dev)
    a, ms_q, min_q = run(model, x0, True)
    b, ms_o, min_o = run(model, x0, False)
    ta_, _, _ = run(model, x0[:1000], True, reps=1, thin=5, return_trajectory=True)
    tb_, _, _ = run(model, x0[:1000], False, reps=1, thin=5, return_trajectory=True)
    print(json.dumps({"dim": dim, "quad_ms": round(ms_q, 4), "quad_min_ms": round(min_q, 4), "old_ms": round(ms_o, 4),
                      "old_min_ms": round(min_o, 4), "speedup": round(ms_o / ms_q, 3),
                      "max_abs_diff": (a - b).abs().max().item(), "finite": bool(torch.isfinite(a).all()),
                      "traj_max_abs_diff": (ta_ - tb_).abs().max().item(), "traj_shape": list(ta_.shape)}), flush=True)
