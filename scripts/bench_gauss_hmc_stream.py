"""HMC on dense Gaussians at dims 164 .. 256 (csrc/gauss_hmc_stream.hip: one ebm_hmc_chain_f32 launch, the slabs of the precision
matrix streamed from the pre-split image): ms per 5 transitions of 10 leapfrog steps at 2^17 chains, beside the same call
without the image (HMC_NO_IMAGE=1: the sampler's per-transition GEMM route)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torchebm_amd as ta
from torchebm_amd import _lib
dev = torch.device("cuda")
def timeit(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
n = 1 << 17
for dim in [int(d) for d in os.environ.get("HMC_DIMS", "164,192,224,256").split(",")]:
    g = torch.Generator().manual_seed(dim)
    a = torch.randn(dim, dim, generator=g)
    model = ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    if os.environ.get("HMC_NO_IMAGE"):
        spec0 = model.fused_spec
        def no_image(spec0=spec0):
            s = spec0(); s.aux = None; return s
        model.fused_spec = no_image
    x = torch.randn(n, dim, device=dev)
    gen = torch.Generator(device=dev).manual_seed(1)
    s = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=10, device=dev)
    c0 = _lib.call_counts["ebm_hmc_chain_f32"]
    ms = timeit(lambda: s.sample(x=x, n_steps=5, generator=gen))
    launches = (_lib.call_counts["ebm_hmc_chain_f32"] - c0) / 4
    out = s.sample(x=x, n_steps=5, generator=gen)
    print(json.dumps({"dim": dim, "n": n, "T": 5, "L": 10, "ms": ms, "hmc_chain_launches_per_call": launches,
                      "useful_TFLOPs": 2 * n * dim * dim * 5 * 11 / ms / 1e9, "finite": bool(torch.isfinite(out).all())}))
