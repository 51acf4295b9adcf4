import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
dev = torch.device("cuda")
def timeit(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
n = 1 << 17
for dim in (160, 192, 224, 256, 254):
    g = torch.Generator().manual_seed(dim)
    a = torch.randn(dim, dim, generator=g)
    model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    x = torch.randn(n, dim, device=dev)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=10, device=dev)
    plain = timeit(lambda: s.sample(x=x, n_steps=5))
    diag = timeit(lambda: s.sample(x=x, n_steps=5, return_diagnostics=True))
    print(json.dumps({"dim": dim, "plain_ms": plain, "with_diagnostics_ms": diag}))
