import json, os, sys, torch
sys.path.insert(0, "/root/repo")
import torchebm_amd as ta
dev = torch.device("cuda")
def timeit(fn, reps=7, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for lg in (16, 18):
    n = 1 << lg
    for K in (8, 16, 32):
        for dim in (20, 21, 24, 28, 30, 32, 33, 60, 62, 64):
            g = torch.Generator().manual_seed(dim)
            m = ta.GaussianMixtureModel(torch.randn(K, dim, generator=g) * 2.0, sigma=1.0, device=dev)
            x = torch.randn(n, dim, device=dev)
            ld = ta.LangevinDynamics(m, step_size=0.01, device=dev)
            hm = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=10, device=dev)
            print(json.dumps({"lg": lg, "K": K, "dim": dim, "langevin_ms": round(timeit(lambda: ld.sample(x=x, n_steps=20)), 3),
                              "hmc_ms": round(timeit(lambda: hm.sample(x=x, n_steps=4)), 3)}), flush=True)
