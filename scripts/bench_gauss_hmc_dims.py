#!/usr/bin/env python3
"""[the EBM_* kernel switches need a library built with make CXXFLAGS_EXTRA=-DEBM_AB_SWITCHES] Gaussian HMC at dims 32..128 (matrix-core kernel; EBM_GAUSS_ROWS=1 forces the lane-group LDS kernel)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
dev = torch.device("cuda")
def timeit(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for dim in [int(d) for d in os.environ.get("DIMS", "16,20,24,32,48,64,96,100,104,108,112,128").split(",")]:
    g = torch.Generator().manual_seed(dim)
    a = torch.randn(dim, dim, generator=g)
    model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    n, T, L = 1 << 16, 10, 10
    x = torch.randn(n, dim, device=dev)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=L, device=dev)
    ms = timeit(lambda: s.sample(x=x, n_steps=T))
    print(json.dumps({"dim": dim, "ms_per_10_transitions": ms, "mh_steps_per_s": n * T / ms * 1e3,
                      "matvec_TFLOPs": n * T * (L + 1) * 2 * dim * dim / ms * 1e3 / 1e12}))
