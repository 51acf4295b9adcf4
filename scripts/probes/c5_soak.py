import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torchebm_amd as ta
from torchebm_amd.utils import GraphedTrainingStep
from torchebm_amd.utils.synthetic import two_moons
dev = torch.device("cuda")
n, k, steps = 65536, 20, 1000
data = two_moons(n, 0.05, seed=0, device=dev)
outs = []
for enabled in (False, True):
    torch.manual_seed(0)
    m = ta.MLPEnergy(2, device=dev)
    s = ta.LangevinDynamics(m, step_size=0.1, noise_scale=1.0, device=dev)
    cd = ta.ContrastiveDivergence(m, s, k_steps=k, persistent=True, buffer_size=n, init_steps=0, device=dev)
    st = GraphedTrainingStep(cd, torch.optim.Adam(m.parameters(), lr=1e-3, capturable=True), generator=torch.Generator(device=dev).manual_seed(1), enabled=enabled)
    torch.cuda.reset_peak_memory_stats()
    losses = torch.stack([st(data)[0] for _ in range(steps)])
    torch.cuda.synchronize()
    outs.append((losses, [p.detach().clone() for p in m.parameters()], cd.replay_buffer.clone()))
    print("enabled", enabled, "final loss", losses[-1].item(), "peak MB", torch.cuda.max_memory_allocated() / 1e6)
print("losses equal", torch.equal(outs[0][0], outs[1][0]), "weights equal", all(torch.equal(a, b) for a, b in zip(outs[0][1], outs[1][1])), "buffer equal", torch.equal(outs[0][2], outs[1][2]))
print("loss trajectory", outs[0][0][::100].tolist())
