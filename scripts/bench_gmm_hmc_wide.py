"""HMC on Gaussian mixtures at 129 .. 256 dims (csrc/gmm_hmc_wide.hip): ms per 4 transitions of 10 leapfrog steps at 2^16 chains,
beside the lane-group kernels (EBM_GMM_ROWS=1 in the environment of a second run; needs a library built with -DEBM_AB_SWITCHES)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchebm_amd as ta
dev = torch.device("cuda")
def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
n = 1 << 16
for K in (8, 16, 32):
    for dim in [int(d) for d in os.environ.get("GMM_DIMS", "128,129,132,160,190,192,200,224,254,256").split(",")]:
        g = torch.Generator().manual_seed(dim)
        m = ta.GaussianMixtureModel(torch.randn(K, dim, generator=g) * 2.0, sigma=1.0, device=dev)
        x = torch.randn(n, dim, device=dev)
        hm = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=10, device=dev)
        ms = timeit(lambda: hm.sample(x=x, n_steps=4))
        print(json.dumps({"K": K, "dim": dim, "n": n, "T": 4, "L": 10, "ms": round(ms, 3)}), flush=True)
