import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torchebm_amd as ta
from torchebm_amd.utils import GraphedTrainingStep
from torchebm_amd.utils.synthetic import two_moons
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
n, k = 65536, 20
data = two_moons(n, 0.05, seed=0, device=dev)
torch.manual_seed(0)
model = ta.MLPEnergy(2, device=dev)
sampler = ta.LangevinDynamics(model, step_size=0.1, noise_scale=1.0, device=dev)
pcd = ta.ContrastiveDivergence(model, sampler, k_steps=k, persistent=True, buffer_size=n, init_steps=0, device=dev)
step = GraphedTrainingStep(pcd, torch.optim.Adam(model.parameters(), lr=1e-3, capturable=True), enabled=False)
for _ in range(3): step(data)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    step(data)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=60, max_name_column_width=60))
