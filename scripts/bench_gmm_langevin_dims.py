#!/usr/bin/env python3
"""[the EBM_* kernel switches need a library built with make CXXFLAGS_EXTRA=-DEBM_AB_SWITCHES] Mixture Langevin over dims and component counts: matrix-layout kernel vs (EBM_GMM_ROWS=1) lane-group kernels.
2^18 chains, k = 50, dense means."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
dev = torch.device("cuda")
def timeit(fn, reps=7, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
n = 1 << 18
for dim in (20, 32, 48, 64, 96, 128):
    x = torch.randn(n, dim, device=dev)
    for K in (4, 8, 16):
        g = torch.Generator().manual_seed(K)
        model = ta.GaussianMixtureModel(torch.randn(K, dim, generator=g) * 2.0, sigma=1.0, device=dev)
        ld = ta.LangevinDynamics(model, step_size=0.01, device=dev)
        ld.donate_input = True
        t = timeit(lambda: ld.sample(x=x.clone(), n_steps=50))
        print(json.dumps({"dim": dim, "K": K, "langevin_ms_k50": t, "frac_of_8TBps": n * 50 * 8 * dim / (t * 1e-3) / 8e12}), flush=True)
