import sys, time, cProfile, pstats, torch
sys.path.insert(0, "/root/repo")
import torchebm_amd as ta
dev = torch.device("cuda")
torch.manual_seed(0)
m = ta.MLPEnergy(2, 128, device=dev)
s = ta.LangevinDynamics(m, step_size=0.05, device=dev)
x0 = torch.randn(65536, 2, device=dev)
def wall(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
def host(fn, reps=50):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    t = (time.perf_counter() - t0) / reps * 1e3
    torch.cuda.synchronize(); return t
f0 = lambda: s.sample(x=x0, n_steps=20)
f1 = lambda: s.sample(x=x0, n_steps=20, thin=5, return_diagnostics=True)
print("pipelined wall ms: plain", wall(f0), "diag", wall(f1))
print("host-only ms: plain", host(f0), "diag", host(f1))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): f1()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
