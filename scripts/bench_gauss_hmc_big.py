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
for dim, n in ((160, 1<<16), (256, 1 << 16), (512, 1 << 15)):
    g = torch.Generator().manual_seed(dim)
    a = torch.randn(dim, dim, generator=g)
    model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    x = torch.randn(n, dim, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=10, device=dev)
    ms_f = timeit(lambda: s.sample(x=x, n_steps=5, generator=gen))
    s._route = lambda xx, kw: ("fused", model.fused_spec())
    ms_s = timeit(lambda: s.sample(x=x, n_steps=5, generator=gen))
    print(json.dumps({"dim": dim, "n": n, "T": 5, "L": 10, "sampler_ms (step route above 128 dims)": ms_f, "fused_kernel_ms": ms_s}))
