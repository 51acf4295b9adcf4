#!/usr/bin/env python3
"""The classic 2-D eight-Gaussians ring at scale: Langevin and HMC throughput (dim 2 is the shape of most of
the reference's examples; at this size the kernels are RNG / latency bound, not bandwidth bound)."""
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
n = 1 << 20
x = torch.randn(n, 2, device=dev)
model = ta.core.ring_mixture(8, 2, device=dev)
ld = ta.LangevinDynamics(model, step_size=0.01, device=dev)
hm = ta.HamiltonianMonteCarlo(model, step_size=0.2, n_leapfrog_steps=10, device=dev)
t_ld = timeit(lambda: ld.sample(x=x, n_steps=100))
t_hm = timeit(lambda: hm.sample(x=x, n_steps=10))
print(json.dumps({"config": "8-Gaussians ring, dim 2, n = 2^20", "langevin_ms_k100": t_ld, "langevin_chain_steps_per_s": n * 100 / t_ld * 1e3,
                  "hmc_ms_T10_L10": t_hm, "hmc_mh_steps_per_s": n * 10 / t_hm * 1e3, "hmc_grad_evals_per_s": n * 10 * 11 / t_hm * 1e3}))
