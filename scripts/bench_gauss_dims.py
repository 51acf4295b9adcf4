import sys, time, json, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients
dev = torch.device('cuda')
def timeit(fn, reps=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
for dim in (3, 5, 8, 12, 16, 20, 24, 30, 32, 48, 50, 64, 96, 100, 128):
    g = torch.Generator().manual_seed(dim)
    a = torch.randn(dim, dim, generator=g)
    model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    k = 50
    for n, route in ((1 << 18, "matrix"), ((1 << 18) + 1, "lane_groups")):
      # one chain more than a multiple of every pack factor: the lane-group kernel takes it (gauss_pack_factor, gauss_mfma.hip)
      if route == "lane_groups" and (dim % 4 == 0 and dim >= 20): continue
      x = torch.randn(n, dim, device=dev)
      spec = model.fused_spec().to_c()
      aa, sq, coef = em_coefficients(0.01, 1.0)
      st = _lib.stream_handle(dev)
      ms = timeit(lambda: _lib.call("ebm_langevin_chain_f32", spec, x.data_ptr(), n, dim, k, aa, sq, coef, None, 0, 0.0, 0.0, 1, None, None, None, 1, 0, st))
      print(json.dumps({"dim": dim, "route": route, "layout": list(_lib.diag_layout(spec, _lib.DIAG_LANGEVIN, n, dim)), "ms": ms, "chain_steps_per_s": n*k/ms*1e3, "frac": n*k*8*dim/ms*1e3/8e12}))
