"""Dense Gaussian Langevin at the widths whose last 16-coordinate K-block is padding (gauss_mfma.hip KT = 1) and their
neighbours: ms per 50 steps x 2^18 chains, step-equivalent fraction of 8 TB/s.  TRIM_DIMS=36,100 restricts the dims."""
import sys, json, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
dims = (36, 40, 48, 50, 52, 64, 68, 72, 80, 84, 96, 100, 104, 112, 116, 128, 132, 144, 148, 160)
if os.environ.get("TRIM_DIMS"): dims = tuple(int(d) for d in os.environ["TRIM_DIMS"].split(","))
for dim in dims:
    g = torch.Generator().manual_seed(dim)
    a = torch.randn(dim, dim, generator=g)
    model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    k, n = 50, 1 << 18
    x = torch.randn(n, dim, device=dev)
    spec = model.fused_spec().to_c()
    aa, sq, coef = em_coefficients(0.01, 1.0)
    st = _lib.stream_handle(dev)
    ms = timeit(lambda: _lib.call("ebm_langevin_chain_f32", spec, x.data_ptr(), n, dim, k, aa, sq, coef, None, 0, 0.0, 0.0, 1, None, None, None, 1, 0, st))
    print(json.dumps({"dim": dim, "ms": round(ms, 4), "chain_steps_per_s": n*k/ms*1e3, "frac": round(n*k*8*dim/ms*1e3/8e12, 4)}))
