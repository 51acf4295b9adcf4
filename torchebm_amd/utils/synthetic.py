"""Synthetic 2-D input for BASELINE config 5 (PCD on two-moons).

Function form of ``torchebm_amd.datasets.TwoMoonsDataset`` (reference: torchebm/datasets/generators.py:272-315):
two half circles, the inner one shifted by (1, -0.5), plus isotropic Gaussian noise; generated on the CPU and moved.
"""

from __future__ import annotations

import math
from typing import Optional, Union

import torch


def two_moons(
    n_samples: int = 2000,
    noise: float = 0.05,
    seed: Optional[int] = None,
    device: Optional[Union[str, torch.device]] = None,
    dtype: torch.dtype = torch.float32,
) -> torch.Tensor:
    n_out = n_samples // 2
    n_in = n_samples - n_out
    t_out = torch.linspace(0, math.pi, n_out, dtype=dtype)
    t_in = torch.linspace(0, math.pi, n_in, dtype=dtype)
    xs = torch.cat((torch.cos(t_out), 1 - torch.cos(t_in)))
    ys = torch.cat((torch.sin(t_out), 1 - torch.sin(t_in) - 0.5))
    data = torch.stack((xs, ys), dim=1)
    gen = None if seed is None else torch.Generator().manual_seed(seed)
    data = data + torch.randn(data.shape, generator=gen, dtype=dtype) * noise
    return data.to(device) if device is not None else data
