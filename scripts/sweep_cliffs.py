"""Cost per element-step of sample() over widths, for every packaged energy and both samplers: ns per (chain x step x
coordinate) -- a cliff between neighbouring widths is a route that has no good kernel.  SWEEP_DIMS / SWEEP_ENERGIES restrict."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
dev = torch.device("cuda")
def timeit(fn, reps=3, warm=1):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return sorted(ts)[len(ts) // 2]
dims = [2, 3, 5, 8, 13, 16, 19, 20, 21, 30, 32, 33, 50, 64, 65, 96, 99, 100, 126, 128, 129, 157, 160, 161, 200, 255, 256, 257, 300, 512, 513, 768, 1024]
if os.environ.get("SWEEP_DIMS"): dims = [int(d) for d in os.environ["SWEEP_DIMS"].split(",")]
energies = os.environ.get("SWEEP_ENERGIES", "dw,gauss,gmm8,gmm16").split(",")
def model_for(kind, dim):
    g = torch.Generator().manual_seed(dim)
    if kind == "dw": return ta.DoubleWellModel(device=dev)
    if kind == "har": return ta.HarmonicModel(device=dev)
    if kind == "gauss":
        a = torch.randn(dim, dim, generator=g)
        return ta.GaussianModel(torch.zeros(dim), a @ a.t() / dim + 0.5 * torch.eye(dim), device=dev)
    k = int(kind[3:])
    return ta.GaussianMixtureModel(torch.randn(k, dim, generator=g) * 2.0, sigma=1.0, device=dev)
n = 1 << 16
for kind in energies:
    for dim in dims:
        try:
            m = model_for(kind, dim)
            x = torch.randn(n, dim, device=dev)
            ld = ta.LangevinDynamics(m, step_size=0.01, device=dev)
            ms_l = timeit(lambda: ld.sample(x=x, n_steps=20))
            hm = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=10, device=dev)
            ms_h = timeit(lambda: hm.sample(x=x, n_steps=4))
            print(json.dumps({"energy": kind, "dim": dim, "langevin_ms": round(ms_l, 3), "langevin_ns_per_elem_step": round(ms_l * 1e6 / (n * 20 * dim), 4),
                              "hmc_ms": round(ms_h, 3), "hmc_ns_per_elem_leapfrog": round(ms_h * 1e6 / (n * 40 * dim), 4)}), flush=True)
        except Exception as ex:  # noqa: BLE001
            print(json.dumps({"energy": kind, "dim": dim, "error": str(ex)[:200]}), flush=True)
