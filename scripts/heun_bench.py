import sys, json, torch
sys.path.insert(0, '.')
import torchebm_amd as ta
dev = torch.device('cuda')
def timeit(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
x = torch.randn(1<<20, 64, device=dev).clamp_(-3, 3)
for integ in (None, "heun"):
    s = ta.LangevinDynamics(ta.DoubleWellModel(device=dev), step_size=0.01, integrator=integ, device=dev)
    print(integ, round(timeit(lambda: s.sample(x=x, n_steps=200)), 3), "ms")
