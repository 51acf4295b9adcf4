"""Which device kernels does ONE sample() call launch besides the chain kernel?  Run under
    rocprofv3 --kernel-trace --stats -- python scripts/host_ops_per_call.py <case>
each case makes exactly 100 calls: a kernel that shows up 100 x (or a multiple) is per-call overhead."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torchebm_amd as ta
dev = torch.device("cuda")
case = sys.argv[1]
n = 4096
g = torch.Generator().manual_seed(0)
if case == "dw_ld":
    s = ta.LangevinDynamics(ta.DoubleWellModel(device=dev), step_size=0.01, device=dev); x = torch.randn(n, 64, device=dev); f = lambda: s.sample(x=x, n_steps=10)
elif case == "gauss_ld":
    a = torch.randn(64, 64, generator=g); m = ta.GaussianModel(torch.zeros(64), a @ a.t() / 64 + 0.5 * torch.eye(64), device=dev)
    s = ta.LangevinDynamics(m, step_size=0.01, device=dev); x = torch.randn(n, 64, device=dev); f = lambda: s.sample(x=x, n_steps=10)
elif case == "gauss_hmc_diag":
    a = torch.randn(64, 64, generator=g); m = ta.GaussianModel(torch.zeros(64), a @ a.t() / 64 + 0.5 * torch.eye(64), device=dev)
    s = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=5, device=dev); x = torch.randn(n, 64, device=dev)
    f = lambda: s.sample(x=x, n_steps=10, thin=5, return_diagnostics=True)
elif case == "gauss200_hmc":
    a = torch.randn(200, 200, generator=g); m = ta.GaussianModel(torch.zeros(200), a @ a.t() / 200 + 0.5 * torch.eye(200), device=dev)
    s = ta.HamiltonianMonteCarlo(m, step_size=0.05, n_leapfrog_steps=5, device=dev); x = torch.randn(n, 200, device=dev); f = lambda: s.sample(x=x, n_steps=4)
elif case == "gmm_hmc":
    s = ta.HamiltonianMonteCarlo(ta.core.ring_mixture(8, 32, device=dev), step_size=0.1, n_leapfrog_steps=5, device=dev); x = torch.randn(n, 32, device=dev); f = lambda: s.sample(x=x, n_steps=4)
elif case == "gmm_ld_diag":
    s = ta.LangevinDynamics(ta.core.ring_mixture(8, 32, device=dev), step_size=0.01, device=dev); x = torch.randn(n, 32, device=dev); f = lambda: s.sample(x=x, n_steps=10, thin=5, return_diagnostics=True)
elif case == "mlp_ld":
    s = ta.LangevinDynamics(ta.MLPEnergy(2, device=dev), step_size=0.01, device=dev); x = torch.randn(n, 2, device=dev); f = lambda: s.sample(x=x, n_steps=10)
for _ in range(100): f()
torch.cuda.synchronize()
