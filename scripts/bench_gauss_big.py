"""Dense Gaussian Langevin at dims above 128 (csrc/gauss_big.hip): ms per call, step-equivalent fraction of 8 TB/s, useful TFLOP/s,
and the same chain as plain torch ops (a GEMM + element-wise ops per step) beside it.  BIG_DIMS=160,256 restricts the dims; BIG_NO_IMAGE=1 runs without the pre-split precision image."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchebm_amd as ta
from torchebm_amd import _lib
from torchebm_amd.samplers.langevin import em_coefficients
dev = torch.device('cuda')
def timeit(fn, reps=5, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
cases = ((132, 1 << 17), (160, 1 << 17), (192, 1 << 17), (224, 1 << 17), (256, 1 << 17), (320, 1 << 16), (384, 1 << 16), (512, 1 << 16), (1024, 1 << 15))
if os.environ.get("BIG_DIMS"):
    want = {int(d) for d in os.environ["BIG_DIMS"].split(",")}
    cases = tuple(c for c in cases if c[0] in want)
elif os.environ.get("BIG_NO_EAGER"):
    cases = tuple(c for c in cases if c[0] in (160, 256, 512))
for dim, n in cases:
    g = torch.Generator().manual_seed(dim)
    a = torch.randn(dim, dim, generator=g)
    model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    k = 20
    x = torch.randn(n, dim, device=dev)
    fs = model.fused_spec()
    spec = fs.to_c()
    if os.environ.get("BIG_NO_IMAGE"):  # A/B: the fp32 slab path (load, split, store every stage)
        spec.aux = None
    aa, sq, coef = em_coefficients(0.01, 1.0)
    st = _lib.stream_handle(dev)
    ms = timeit(lambda: _lib.call("ebm_langevin_chain_f32", spec, x.data_ptr(), n, dim, k, aa, sq, coef, None, 0, 0.0, 0.0, 1, None, None, None, 1, 0, st))
    row = {"dim": dim, "n": n, "k": k, "ms": ms, "frac": n * k * 8 * dim / ms * 1e3 / 8e12, "useful_TFLOPs": 2 * n * k * dim * dim / ms * 1e3 / 1e12}
    if not os.environ.get("BIG_NO_EAGER"):
        xx = torch.randn(n, dim, device=dev); Pm = torch.randn(dim, dim, device=dev)
        def eager():
            y = xx
            for _ in range(k):
                y = y - 0.01 * (y @ Pm) + 0.1 * torch.randn_like(y)
            return y
        row["torch_eager_ms"] = timeit(eager)
    print(json.dumps(row))
# widths without a matrix-core chain kernel: the sampler's step route (library GEMM + fused update, replayed from a HIP graph)
if not os.environ.get("BIG_DIMS") and not os.environ.get("BIG_NO_EAGER"):
    for dim, n in ((130, 1 << 17), (1024, 1 << 15)):
        g = torch.Generator().manual_seed(dim)
        a = torch.randn(dim, dim, generator=g)
        model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
        s = ta.LangevinDynamics(model, step_size=0.01, device=dev)
        x = torch.randn(n, dim, device=dev)
        gen = torch.Generator(device=dev).manual_seed(1)
        ms = timeit(lambda: s.sample(x=x, n_steps=20, generator=gen), reps=5, warm=2)
        print(json.dumps({"dim": dim, "n": n, "k": 20, "route": "sampler: GEMM step route", "ms": ms, "frac": n * 20 * 8 * dim / ms * 1e3 / 8e12}))
