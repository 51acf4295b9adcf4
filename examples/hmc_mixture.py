"""HMC on an 8-mode Gaussian mixture ring (BASELINE config 3's energy)."""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout

from torchebm_amd.core import ring_mixture
from torchebm_amd.samplers import HamiltonianMonteCarlo

SMOKE = os.getenv("TORCHEBM_SMOKE") == "1"
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

energy = ring_mixture(n_components=8, dim=2, radius=4.0, sigma=1.0, device=device)
sampler = HamiltonianMonteCarlo(energy, step_size=0.2, n_leapfrog_steps=10, device=device)
n, steps = (128, 10) if SMOKE else (50_000, 300)
x, diag = sampler.sample(n_samples=n, n_steps=steps, thin=max(1, steps // 5), return_diagnostics=True)
print(f"device={device}  acceptance rate per kept step: {[round(v, 3) for v in diag['acceptance_rate'].tolist()]}")
angle = torch.atan2(x[:, 1], x[:, 0])
mode = torch.round(angle / (2 * torch.pi / 8)).long() % 8
print("chains per mode:", torch.bincount(mode, minlength=8).tolist(), " mean radius:", round(x.norm(dim=1).mean().item(), 3))
