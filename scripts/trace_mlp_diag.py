import sys, torch
sys.path.insert(0, "/root/repo")
import torchebm_amd as ta
dev = torch.device("cuda")
torch.manual_seed(0)
m = ta.MLPEnergy(2, 128, device=dev)
s = ta.LangevinDynamics(m, step_size=0.05, device=dev)
x0 = torch.randn(65536, 2, device=dev)
mode = sys.argv[1]
for _ in range(100):
    if mode == "diag": s.sample(x=x0, n_steps=20, thin=5, return_diagnostics=True)
    else: s.sample(x=x0, n_steps=20)
torch.cuda.synchronize()
