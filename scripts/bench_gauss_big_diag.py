import sys, json, torch
sys.path.insert(0, "/root/repo")
import torchebm_amd as ta
dev = torch.device("cuda")
def timeit(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
dim, n = 256, 1 << 17
g = torch.Generator().manual_seed(dim); a = torch.randn(dim, dim, generator=g)
model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
s = ta.LangevinDynamics(model, step_size=0.01, device=dev)
x = torch.randn(n, dim, device=dev)
print(json.dumps({"plain_ms": timeit(lambda: s.sample(x=x, n_steps=20)), "diag_thin5_ms": timeit(lambda: s.sample(x=x, n_steps=20, thin=5, return_diagnostics=True))}))
