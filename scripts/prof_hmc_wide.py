#!/usr/bin/env python3
"""Where a wide-MLP HMC transition's time goes on the per-transition route: the pieces, timed with events."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
from torchebm_amd import _lib
dev = torch.device("cuda")
torch.manual_seed(0)
dim, hidden = int(os.environ.get("DIM", 8)), int(os.environ.get("HID", 128))
m = ta.MLPEnergy(dim, hidden, device=dev)
x = torch.randn(65536, dim, device=dev)


def ev(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


print("gradient()            ms", ev(lambda: m.gradient(x)))
spec = m.fused_spec()
g = torch.empty_like(x)
c = spec.to_c()
print("ebm_energy_grad only  ms", ev(lambda: _lib.call("ebm_energy_grad_f32", c, x.data_ptr(), x.shape[0], dim, None, g.data_ptr(), _lib.stream_handle(dev))))
print("fused_spec()          ms", ev(lambda: m.fused_spec()))
with torch.no_grad():
    print("forward (energy)      ms", ev(lambda: m(x)))
s = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=10, device=dev)
s.capture_graph = False
print("sample 10 transitions ms", ev(lambda: s.sample(x=x, n_steps=10), reps=3))
