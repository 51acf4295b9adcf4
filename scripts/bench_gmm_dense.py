#!/usr/bin/env python3
"""[the EBM_* kernel switches need a library built with make CXXFLAGS_EXTRA=-DEBM_AB_SWITCHES] HMC on mixtures with DENSE means (every column differs): the matrix-layout kernel against the lane-group kernels
(EBM_GMM_ROWS=1), K = 4 .. 32, dim 32 / 64, 2^18 chains, L = 20, 10 transitions per call."""
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
for dim in (32, 64, 128):
    x = torch.randn(n, dim, device=dev)
    for K in (4, 8, 16, 32):
        g = torch.Generator().manual_seed(K)
        model = ta.GaussianMixtureModel(torch.randn(K, dim, generator=g) * 2.0, sigma=1.0, device=dev)
        hm = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=20, device=dev)
        hm.donate_input = True
        t = timeit(lambda: hm.sample(x=x.clone(), n_steps=10))
        print(json.dumps({"dim": dim, "K": K, "hmc_ms_T10_L20": t, "mh_steps_per_s": n * 10 / t * 1e3}), flush=True)
