import time, torch, sys
sys.path.insert(0, '/root/repo')
import torchebm_amd as ta
dev = torch.device('cuda')
def wall(fn, reps=200, warm=20):
    for _ in range(warm): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
x = torch.randn(1024, 2, device=dev)
for name, model in [("dw", ta.DoubleWellModel(device=dev)), ("gauss", ta.GaussianModel(torch.zeros(2), torch.eye(2), device=dev)),
                    ("gmm", ta.core.ring_mixture(4, 2, device=dev)), ("mlp", ta.MLPEnergy(2, device=dev))]:
    s = ta.LangevinDynamics(model, step_size=0.01, device=dev)
    print(name, "langevin sample() us/call:", round(wall(lambda: s.sample(x=x, n_steps=10)) * 1e6, 1))
    if name != "mlp":
        h = ta.HamiltonianMonteCarlo(model, step_size=0.05, n_leapfrog_steps=5, device=dev)
        print(name, "hmc sample() us/call:", round(wall(lambda: h.sample(x=x, n_steps=4)) * 1e6, 1))
