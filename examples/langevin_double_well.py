"""Langevin sampling of the double-well energy (cf. the reference's examples/10-sampling/01-mcmc).

Identical to the reference script except for the import line: on a CUDA device the whole k-step
loop is one fused HIP kernel launch."""

import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # run from a source checkout

from torchebm_amd.core import DoubleWellModel
from torchebm_amd.samplers import LangevinDynamics

SMOKE = os.getenv("TORCHEBM_SMOKE") == "1"
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")

energy = DoubleWellModel(barrier_height=2.0, device=device)
sampler = LangevinDynamics(energy, step_size=0.01, noise_scale=1.0, device=device)
n, k = (256, 50) if SMOKE else (100_000, 1000)
samples, diag = sampler.sample(n_samples=n, dim=2, n_steps=k, thin=max(1, k // 10), return_diagnostics=True)
print(f"device={device}  samples {tuple(samples.shape)}  E|x| = {samples.abs().mean().item():.3f} (stationary: 0.868)")
print("mean energy along the run:", [round(v, 3) for v in diag["energy"].tolist()])
left = (samples[:, 0] < 0).float().mean().item()
print(f"fraction in the left well (x0 < 0): {left:.3f}")
