#!/usr/bin/env python3
"""Mixture size sweep (K = 4..32, dim 32): K <= 8 takes the scalar-operand / padded small-mixture paths,
larger K the online-softmax loop."""
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
n, dim = 1 << 18, 32
x = torch.randn(n, dim, device=dev)
for K in (4, 8, 9, 16, 25, 32):
    model = ta.core.ring_mixture(K, dim, device=dev)
    ld = ta.LangevinDynamics(model, step_size=0.01, device=dev)
    hm = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=20, device=dev)
    t_ld = timeit(lambda: ld.sample(x=x, n_steps=50))
    t_hm = timeit(lambda: hm.sample(x=x, n_steps=10))
    print(json.dumps({"K": K, "langevin_ms_k50": t_ld, "langevin_chain_steps_per_s": n * 50 / t_ld * 1e3,
                      "hmc_ms_T10_L20": t_hm, "hmc_mh_steps_per_s": n * 10 / t_hm * 1e3}))
