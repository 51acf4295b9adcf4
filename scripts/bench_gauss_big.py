import sys, json, torch, os
sys.path.insert(0, "/root/repo")
import torchebm_amd as ta
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients
dev = torch.device('cuda')
def timeit(fn, reps=5, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
for dim, n in ((160, 1<<17), (192, 1 << 17), (256, 1 << 17), (512, 1 << 16), (1024, 1 << 15)):
    g = torch.Generator().manual_seed(dim)
    a = torch.randn(dim, dim, generator=g)
    model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    k = 20
    x = torch.randn(n, dim, device=dev)
    spec = model.fused_spec().to_c()
    aa, sq, coef = em_coefficients(0.01, 1.0)
    st = _lib.stream_handle(dev)
    ms = timeit(lambda: _lib.call("ebm_langevin_chain_f32", spec, x.data_ptr(), n, dim, k, aa, sq, coef, None, 0, 0.0, 0.0, 1, None, None, None, 1, 0, st))
    # torch eager: k steps of x - eta * (x-mu)@P + noise
    P = model.precision if hasattr(model, "precision") else None
    print(json.dumps({"dim": dim, "n": n, "k": k, "ms": ms, "frac": n*k*8*dim/ms*1e3/8e12, "useful_TFLOPs": 2*n*k*dim*dim/ms*1e3/1e12}))
    xx = torch.randn(n, dim, device=dev); Pm = torch.randn(dim, dim, device=dev)
    def eager():
        y = xx
        for _ in range(k):
            y = y - 0.01 * (y @ Pm) + 0.1 * torch.randn_like(y)
        return y
    ms2 = timeit(eager)
    print(json.dumps({"dim": dim, "torch_eager_ms": ms2, "frac": n*k*8*dim/ms2*1e3/8e12}))
