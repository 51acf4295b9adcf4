"""Oracle: noise-free descent samplers (torchebm/samplers/gradient_descent.py:119-131, :258-274).

Written with the reference's own torch calls (``torch.sub/add(..., alpha=)``, ``mul_().sub_()``) so
that the rounding (single-rounding FMAs in ATen's CPU kernels) is reproduced, not re-derived."""

from __future__ import annotations

from typing import Optional, Sequence

import torch


def descent_chain(energy, x0: torch.Tensor, etas: Sequence[float], momentum: Optional[float] = None, thin: int = 1,
                  want_traj: bool = False, want_diag: bool = False):
    """``momentum=None``: gradient descent; otherwise Nesterov with that coefficient.
    Returns ``(x, trajectory_or_None, energy_diagnostics_or_None)``."""
    x = x0.clone()
    k = len(etas)
    n_kept = k // thin
    traj = torch.empty((x.shape[0], n_kept) + tuple(x.shape[1:]), dtype=x.dtype) if want_traj else None
    en_diag = torch.empty(n_kept, dtype=x.dtype) if want_diag else None
    v = torch.zeros_like(x) if momentum is not None else None
    keep = 0
    for i in range(k):
        if v is None:
            x = torch.sub(x, energy.grad(x), alpha=etas[i])                 # :121-123
        else:
            lookahead = torch.add(x, v, alpha=momentum)                     # :262
            v.mul_(momentum).sub_(energy.grad(lookahead), alpha=etas[i])    # :264
            x = x + v                                                       # :266
        if (i + 1) % thin == 0:
            if traj is not None:
                traj[:, keep] = x
            if en_diag is not None:
                en_diag[keep] = energy.energy(x).mean()
            keep += 1
    return x, traj, en_diag
