import sys, json, torch
sys.path.insert(0, "/root/repo")
import torchebm_amd as ta
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients
dev = torch.device("cuda")
def timeit(fn, reps=40, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(reps):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts)//2]
for dim in (2, 32):
    torch.manual_seed(0)
    m = ta.MLPEnergy(dim, 128, device=dev)
    n, k = 65536, 20
    x = torch.randn(n, dim, device=dev)
    spec = m.fused_spec().to_c()
    a, sq, coef = em_coefficients(0.05, 1.0)
    st = _lib.stream_handle(dev)
    out = {}
    for thin in (5, 0, 20, 7, 5, 3, 1, 0):  # (the first case of a process runs on a cold clock: measured twice, first one discarded)
        rec = None
        if thin:
            nb, S, E = _lib.diag_layout(spec, _lib.DIAG_LANGEVIN, n, dim, False, False)
            rec = torch.empty((k // thin) * nb * (2 * S + 8), device=dev)
        t = thin or 1
        out[f"thin{thin}"] = timeit(lambda: _lib.call("ebm_langevin_chain_f32", spec, x.data_ptr(), n, dim, k, a, sq, coef, None, 0, 0.0, 0.0, t, None, _lib.ptr(rec), None, 1, 0, st))
    print(json.dumps({"dim": dim, **out}))
