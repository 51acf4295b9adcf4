import json, os, sys, torch
sys.path.insert(0, "/root/repo")
import torchebm_amd as ta
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients
dev = torch.device("cuda")
def timeit(fn, reps=5, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
for dim in (161, 190, 254):
    n = 1 << 17
    g = torch.Generator().manual_seed(dim)
    a = torch.randn(dim, dim, generator=g)
    model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    x = torch.randn(n, dim, device=dev)
    spec = model.fused_spec().to_c()
    aa, sq, coef = em_coefficients(0.01, 1.0)
    st = _lib.stream_handle(dev)
    ms = timeit(lambda: _lib.call("ebm_langevin_chain_f32", spec, x.data_ptr(), n, dim, 20, aa, sq, coef, None, 0, 0.0, 0.0, 1, None, None, None, 1, 0, st))
    print(json.dumps({"dim": dim, "n": n, "k": 20, "ms": ms}))
