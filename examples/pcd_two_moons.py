"""Persistent contrastive divergence on two-moons (cf. the reference's
examples/20-training/01-mcmc-losses/02-persistent-cd/main.py; BASELINE config 5).

The energy is the reference example's 2-128-128-1 SiLU MLP.  Defined by hand (as in the reference
script) the sampler uses autograd for the gradient and one fused HIP launch per Langevin step;
with the packaged `MLPEnergy` (same network) all k steps -- forward, input-gradient on the matrix
cores, update, noise -- are ONE kernel launch.  Set TORCHEBM_HANDWRITTEN_MLP=1 for the former;
TORCHEBM_CAPTURE_GRAPH=1 additionally replays its per-step loop from a HIP graph.
TORCHEBM_WHOLE_STEP_GRAPH=1 (packaged energy, CUDA): the WHOLE training step -- start points, chain, buffer write, loss, backward,
Adam -- replays from one HIP graph (utils.GraphedTrainingStep): same losses and weights as the loop below, bit for bit."""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout
from torch import nn

from torchebm_amd.core import BaseModel, MLPEnergy as FusedMLPEnergy
from torchebm_amd.losses import ContrastiveDivergence
from torchebm_amd.samplers import LangevinDynamics
from torchebm_amd.utils.synthetic import two_moons

SMOKE = os.getenv("TORCHEBM_SMOKE") == "1"
N_STEPS = 20 if SMOKE else 1000
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class MLPEnergy(BaseModel):
    def __init__(self):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(2, 128), nn.SiLU(), nn.Linear(128, 128), nn.SiLU(), nn.Linear(128, 1))

    def forward(self, x):
        return self.net(x).squeeze(-1)


torch.manual_seed(0)
data = two_moons(n_samples=3000, noise=0.05, seed=0, device=device)
energy = MLPEnergy().to(device) if os.getenv("TORCHEBM_HANDWRITTEN_MLP") == "1" else FusedMLPEnergy(2, device=device)
sampler = LangevinDynamics(model=energy, step_size=0.1, noise_scale=1.0, device=device)
sampler.capture_graph = device.type == "cuda" and os.getenv("TORCHEBM_CAPTURE_GRAPH") == "1"  # step route only
pcd = ContrastiveDivergence(model=energy, sampler=sampler, k_steps=10, persistent=True, buffer_size=8192, device=device)
whole_step = device.type == "cuda" and os.getenv("TORCHEBM_WHOLE_STEP_GRAPH") == "1" and isinstance(energy, FusedMLPEnergy)
opt = torch.optim.Adam(energy.parameters(), lr=1e-3, capturable=whole_step)
if whole_step:
    from torchebm_amd.utils import GraphedTrainingStep

    graphed = GraphedTrainingStep(pcd, opt)  # two eager steps, then one graph launch per step

for step in range(N_STEPS):
    batch = data[torch.randint(len(data), (256,), device=device)]
    if whole_step:
        loss, negatives = graphed(batch)
    else:
        loss, negatives = pcd(batch)
        opt.zero_grad()
        loss.backward()
        opt.step()
    if step % 200 == 0 or step == N_STEPS - 1:
        with torch.no_grad():  # (a diagnostic: no autograd graph -- and none that outlives the step, which a captured step could not share)
            gap = energy(negatives).mean() - energy(batch).mean()
        print(f"step {step:4d}  loss {loss.item():+.3f}  E(neg) - E(data) = {gap.item():+.3f}")
print("done on", device)
