"""The two synthetic inputs either side of the hot path (reference package: torchebm/datasets).

* ``TwoMoonsDataset``         -- the data of BASELINE config 5 (PCD training of the 2-128-128-1 MLP energy;
                                 examples/20-training/01-mcmc-losses of the reference import it);
* ``GaussianMixtureDataset``  -- K modes on a ring, whose centres are the component means of config 3's mixture
                                 energy (``core.ring_mixture``).

The reference's other generators (swiss roll, checkerboard, pinwheel, ...) feed its flow / score-matching examples and
are outside SURVEY.md §8.  For a given ``seed`` the tensors here are the reference's (generators.py:176-202, :295-315:
same draw order from a generator seeded the same way -- ``tests/test_host_api.py`` compares them where the reference is
present); unlike the reference a seeded dataset does not reseed torch's GLOBAL generators as a side effect.
"""

from __future__ import annotations

import math
import warnings
from typing import Optional, Union

import torch
from torch.utils.data import Dataset

__all__ = ["GaussianMixtureDataset", "TwoMoonsDataset"]


class _PointCloud(Dataset):
    """``n_samples`` points generated once at construction and kept as one tensor (``get_data``); indexable as a
    ``torch.utils.data.Dataset``; ``regenerate(seed)`` draws again."""

    def __init__(self, n_samples: int, device=None, dtype: torch.dtype = torch.float32, seed: Optional[int] = None):
        if n_samples <= 0:
            raise ValueError("n_samples must be positive")
        self.n_samples, self.device, self.dtype, self.seed = n_samples, device, dtype, seed
        self.data: Optional[torch.Tensor] = None
        self._build()

    def _draw(self, gen: Optional[torch.Generator]) -> torch.Tensor:  # pragma: no cover - abstract
        raise NotImplementedError

    def _gen(self) -> Optional[torch.Generator]:
        if self.seed is None:
            return None
        dev = torch.device(self.device) if self.device is not None else torch.device("cpu")
        return torch.Generator(device=dev if dev.type == "cuda" else "cpu").manual_seed(int(self.seed))

    def _build(self) -> None:
        points = self._draw(self._gen())
        self.data = torch.as_tensor(points).to(dtype=self.dtype, device=self.device)
        if self.data.shape[0] != self.n_samples:
            warnings.warn(f"generated {self.data.shape[0]} samples where {self.n_samples} were requested", RuntimeWarning)

    def regenerate(self, seed: Optional[int] = None) -> None:
        if seed is not None:
            self.seed = seed
        self._build()

    def get_data(self) -> torch.Tensor:
        if self.data is None:
            self._build()
        return self.data

    def __len__(self) -> int:
        return self.n_samples

    def __getitem__(self, idx: int) -> torch.Tensor:
        if not 0 <= idx < self.n_samples:
            raise IndexError(f"Index {idx} out of bounds for dataset with size {self.n_samples}")
        return self.get_data()[idx]

    def __repr__(self) -> str:
        extra = ", ".join(f"{k}={v}" for k, v in vars(self).items() if k not in ("data", "n_samples", "device", "dtype", "seed"))
        return f"{type(self).__name__}(n_samples={self.n_samples}, {extra + ', ' if extra else ''}seed={self.seed})"


class TwoMoonsDataset(_PointCloud):
    """Two interleaved half circles with isotropic Gaussian jitter: the upper arc on the unit circle, the lower one
    mirrored and shifted by (1, -0.5)."""

    def __init__(self, n_samples: int = 2000, noise: float = 0.05, device=None, dtype: torch.dtype = torch.float32,
                 seed: Optional[int] = None):
        self.noise = noise
        super().__init__(n_samples, device, dtype, seed)

    def _draw(self, gen):
        n_upper = self.n_samples // 2
        upper = torch.linspace(0, math.pi, n_upper, device=self.device, dtype=self.dtype)
        lower = torch.linspace(0, math.pi, self.n_samples - n_upper, device=self.device, dtype=self.dtype)
        xs = torch.cat((torch.cos(upper), 1 - torch.cos(lower)))
        ys = torch.cat((torch.sin(upper), 1 - torch.sin(lower) - 0.5))
        pts = torch.stack((xs, ys), dim=1)
        return pts + torch.randn(pts.shape, generator=gen, device=pts.device, dtype=pts.dtype) * self.noise


class GaussianMixtureDataset(_PointCloud):
    """``n_components`` isotropic Gaussians of width ``std`` whose centres are evenly spaced on a circle of ``radius``;
    the samples are dealt to the components as evenly as ``n_samples`` allows and shuffled."""

    def __init__(self, n_samples: int = 2000, n_components: int = 8, std: float = 0.05, radius: float = 1.0, device=None,
                 dtype: torch.dtype = torch.float32, seed: Optional[int] = None):
        if n_components <= 0:
            raise ValueError("n_components must be positive")
        if std < 0:
            raise ValueError("std must be non-negative")
        self.n_components, self.std, self.radius = n_components, std, radius
        super().__init__(n_samples, device, dtype, seed)

    def centers(self) -> torch.Tensor:
        ang = torch.linspace(0, 2 * math.pi, self.n_components + 1, device=self.device, dtype=self.dtype)[:-1]
        return torch.stack((self.radius * torch.cos(ang), self.radius * torch.sin(ang)), dim=1)

    def _draw(self, gen):
        centres = self.centers()
        share, extra = divmod(self.n_samples, self.n_components)
        blocks = []
        for i in range(self.n_components):
            count = share + (1 if i < extra else 0)
            if count:
                jitter = torch.randn(count, 2, generator=gen, device=centres.device, dtype=centres.dtype) * self.std
                blocks.append(centres[i] + jitter)
        pts = torch.cat(blocks)
        return pts[torch.randperm(self.n_samples, generator=gen, device=pts.device)]
