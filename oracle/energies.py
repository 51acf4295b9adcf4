"""Oracle energies: ``energy(x) -> [B]`` and ``grad(x) -> [B, dim]`` on CPU tensors.

The reference computes every gradient with autograd (torchebm/core/base_model.py:62-127).
For the element-wise energies the closed forms below are written in autograd's operation
order and are bit-identical to it on CPU (checked in tests/test_oracle_golden.py); for the
matrix/mixture energies the oracle simply runs autograd on the restated forward, which is
what the reference itself does.
"""

from __future__ import annotations

import torch


def _autograd(energy_fn, x: torch.Tensor) -> torch.Tensor:
    # base_model.py:93-124: detach, requires_grad, forward, grad with ones
    with torch.enable_grad():
        leaf = x.detach().clone().requires_grad_(True)
        e = energy_fn(leaf)
        (g,) = torch.autograd.grad(e, leaf, grad_outputs=torch.ones_like(e))
    return g.detach()


class DoubleWell:
    """base_model.py:130-148: E = h * sum((x^2 - b^2)^2)."""

    def __init__(self, h: float = 2.0, b: float = 1.0):
        self.h, self.b = h, b

    def energy(self, x):
        return self.h * (x.pow(2) - self.b**2).pow(2).sum(dim=-1)

    def grad(self, x):
        u = x.pow(2) - self.b**2
        return (self.h * (2.0 * u)) * (2.0 * x)  # two pow-backward nodes, in this order

    def grad_autograd(self, x):
        return _autograd(self.energy, x)


class Harmonic:
    """base_model.py:213-229: E = 0.5 * k * sum(x^2)."""

    def __init__(self, k: float = 1.0):
        self.k = k

    def energy(self, x):
        return 0.5 * self.k * x.pow(2).sum(dim=-1)

    def grad(self, x):
        return (0.5 * self.k) * (2.0 * x)

    def grad_autograd(self, x):
        return _autograd(self.energy, x)


class Gaussian:
    """base_model.py:151-210: E = 0.5 d^T P d with P = inverse(cov), bmm form for B > 1."""

    def __init__(self, mean: torch.Tensor, cov: torch.Tensor):
        self.mean = mean.to(torch.float32)
        self.cov_inv = torch.inverse(cov).to(torch.float32)  # :172

    def energy(self, x):
        delta = x - self.mean
        if delta.shape[0] > 1:  # :199-206
            pexp = self.cov_inv.unsqueeze(0).expand(delta.shape[0], -1, -1)
            temp = torch.bmm(pexp, delta.unsqueeze(-1))
            return 0.5 * torch.bmm(delta.unsqueeze(1), temp).squeeze(-1).squeeze(-1)
        return 0.5 * torch.sum(delta * torch.matmul(delta, self.cov_inv), dim=-1)  # :208

    def grad(self, x):
        return _autograd(self.energy, x)

    grad_autograd = grad


class GaussianMixture:
    """Not in the reference: E = -logsumexp_k(log w_k - |x-mu_k|^2/(2 s^2)).
    The oracle is the reference's sampler loop driving autograd on this forward."""

    def __init__(self, means: torch.Tensor, sigma: float = 1.0, log_weights: torch.Tensor = None):
        self.means = means.to(torch.float32)
        self.sigma = float(sigma)
        k = means.shape[0]
        self.log_weights = (
            torch.full((k,), -torch.log(torch.tensor(float(k))).item(), dtype=torch.float32)
            if log_weights is None
            else log_weights.to(torch.float32)
        )

    def energy(self, x):
        sq = (x.unsqueeze(1) - self.means.unsqueeze(0)).pow(2).sum(dim=-1)
        return -torch.logsumexp(self.log_weights - sq / (2.0 * self.sigma**2), dim=1)

    def grad(self, x):
        return _autograd(self.energy, x)

    grad_autograd = grad
