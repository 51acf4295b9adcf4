#!/usr/bin/env python3
"""Noise-free descent samplers (gradient descent / Nesterov) on the config-2 shape."""
import sys, os, json, torch
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
n, dim, k = 1 << 20, 64, 200
x = torch.randn(n, dim, device=dev).clamp_(-3, 3)
model = ta.DoubleWellModel(device=dev)
gd = ta.samplers.GradientDescentSampler(model, step_size=0.01, device=dev)
nag = ta.samplers.NesterovSampler(model, step_size=0.01, momentum=0.9, device=dev)
for name, s in (("gradient_descent", gd), ("nesterov", nag)):
    ms = timeit(lambda: s.sample(x=x, n_steps=k))
    print(json.dumps({"sampler": name, "ms": ms, "chain_steps_per_s": n * k / ms * 1e3, "step_equiv_frac_of_8TBps": n * k * 8 * dim / ms * 1e3 / 8e12}))
