#!/usr/bin/env python3
"""Cost of return_diagnostics=True on the fused routes (in-kernel records, include/ebm_hip.h: diag_partials):
kernel time of the chain launch with and without records, plus the merge kernel, for BASELINE configs 2 and 3."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torchebm_amd as ta  # noqa: E402
from torchebm_amd import _lib  # noqa: E402

dev = torch.device("cuda")


def kernel_ms(entries, fn, reps=5, warm=2):
    """Median over `reps` calls of the summed GPU time of each entry point's launches within one call."""
    for _ in range(warm):
        fn()
    per_call = {e: [] for e in entries}
    for _ in range(reps):
        for e in entries:
            _lib.timed_events[e] = []
        fn()
        torch.cuda.synchronize()
        for e in entries:
            per_call[e].append(sum(a.elapsed_time(b) for a, b in _lib.timed_events.pop(e)))
    return {e: sorted(v)[len(v) // 2] for e, v in per_call.items()}


n, dim, k = 1 << 20, 64, 200
s = ta.LangevinDynamics(ta.DoubleWellModel(device=dev), step_size=0.01, device=dev)
x0 = torch.randn(n, dim, device=dev).clamp_(-4, 4)
gen = torch.Generator(device=dev).manual_seed(1)
ents = ["ebm_langevin_chain_f32", "ebm_diag_finish_f32"]
base = kernel_ms(ents, lambda: s.sample(x=x0, n_steps=k, generator=gen))
for thin in (50, 10, 1):
    d = kernel_ms(ents, lambda: s.sample(x=x0, n_steps=k, thin=thin, return_diagnostics=True, generator=gen), reps=5, warm=2)
    print(json.dumps({"case": f"config2 Langevin DoubleWell 2^20x64 k=200, return_diagnostics thin={thin}",
                      "chain_ms_plain": base[ents[0]], "chain_ms_with_records": d[ents[0]], "merge_ms": d[ents[1]],
                      "ratio": d[ents[0]] / base[ents[0]], "kept_steps": k // thin}), flush=True)

n, dim, T, L = 1 << 18, 32, 50, 20
h = ta.HamiltonianMonteCarlo(ta.core.ring_mixture(8, dim, device=dev), step_size=0.1, n_leapfrog_steps=L, device=dev)
x0 = torch.randn(n, dim, device=dev)
ents = ["ebm_hmc_chain_f32", "ebm_diag_finish_f32"]
base = kernel_ms(ents, lambda: h.sample(x=x0, n_steps=T, generator=gen))
for thin in (50, 5, 1):
    d = kernel_ms(ents, lambda: h.sample(x=x0, n_steps=T, thin=thin, return_diagnostics=True, generator=gen), reps=5, warm=2)
    print(json.dumps({"case": f"config3 HMC GMM-8 2^18x32 L=20 T=50, return_diagnostics thin={thin}",
                      "chain_ms_plain": base[ents[0]], "chain_ms_with_records": d[ents[0]], "merge_ms": d[ents[1]],
                      "ratio": d[ents[0]] / base[ents[0]], "kept_steps": T // thin}), flush=True)
