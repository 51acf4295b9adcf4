import sys, json, torch
sys.path.insert(0, "/root/repo")
import torchebm_amd as ta
dev = torch.device("cuda")
def timeit(fn, reps=5, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
n = 1 << 22
x = torch.randn(n, 2, device=dev)
for name, model in (("gauss2", ta.GaussianModel(torch.zeros(2), torch.tensor([[1.0, 0.8], [0.8, 1.0]]), device=dev)), ("gmm8", ta.core.ring_mixture(8, 2, device=dev)), ("dw", ta.DoubleWellModel(device=dev))):
    for L in (5, 20):
        h = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=L, device=dev)
        ms = timeit(lambda: h.sample(x=x, n_steps=10))
        print(json.dumps({"model": name, "L": L, "ms_per_10_transitions": ms, "mh_steps_per_s": n * 10 / ms * 1e3}))
    s = ta.LangevinDynamics(model, step_size=0.01, device=dev)
    print(json.dumps({"model": name, "langevin_100_steps_ms": timeit(lambda: s.sample(x=x, n_steps=100))}))
