import sys, json, torch
sys.path.insert(0, '.')
import torchebm_amd as ta
dev = torch.device('cuda')
def timeit(fn, reps=6, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
n, dim = 1<<18, 32
x = torch.randn(n, dim, device=dev)
model = ta.core.ring_mixture(8, dim, device=dev)
for mass in (None, 2.0, torch.rand(dim, device=dev) + 0.5):
    h = ta.HamiltonianMonteCarlo(model, step_size=0.1, n_leapfrog_steps=20, mass=mass, device=dev)
    ms = timeit(lambda: h.sample(x=x, n_steps=10))
    print("mass", "none" if mass is None else ("scalar" if isinstance(mass, float) else "diag"), round(ms, 3), "ms per 10 transitions")
