"""Oracle: leapfrog and the HMC transition loop with injected momentum / uniform draws.

Follows torchebm/integrators/leapfrog.py:116-187 (integrate, safe mode),
torchebm/core/base_integrator.py:855-889 (step-size tensorisation, clamp, NaN scrub) and
torchebm/samplers/hmc.py:92-159 (momentum, kinetic energy), :243-312 (loop).
"""

from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Union

import torch

Mass = Optional[Union[float, torch.Tensor]]


def leapfrog(energy, x: torch.Tensor, p: torch.Tensor, eps: float, n_steps: int, mass: Mass = None, safe: bool = True):
    """leapfrog.py:156-185.  ``eps`` becomes an fp32 0-d tensor first (base_integrator.py:869-870),
    so ``0.5 * eps_t`` is an fp32 scalar multiplied into the force."""
    eps_t = torch.tensor(eps, dtype=x.dtype)
    for _ in range(n_steps):
        force = -energy.grad(x)
        if safe:
            force.clamp_(min=-1e6, max=1e6)
        p_half = p + 0.5 * eps_t * force
        if mass is None:
            x = x + eps_t * p_half
        elif isinstance(mass, float):
            x = x + eps_t * p_half / max(mass, 1e-10)
        else:
            x = x + eps_t * p_half / torch.clamp(mass, min=1e-10).view((1,) * (x.ndim - 1) + (-1,))
        force_new = -energy.grad(x)
        if safe:
            force_new.clamp_(min=-1e6, max=1e6)
        p = p_half + 0.5 * eps_t * force_new
        if safe:
            x.nan_to_num_(nan=0.0)
            p.nan_to_num_(nan=0.0)
    return x, p


def kinetic(p: torch.Tensor, mass: Mass) -> torch.Tensor:
    """hmc.py:136-159."""
    if mass is None:
        return 0.5 * torch.sum(p.square(), dim=-1)
    if isinstance(mass, float):
        return 0.5 * torch.sum(p.square(), dim=-1) / mass
    return 0.5 * torch.sum(p.square() / mass.view((1,) * (p.ndim - 1) + (-1,)), dim=-1)


def hmc_chain(
    energy,
    x0: torch.Tensor,
    p_noise: torch.Tensor,
    u: torch.Tensor,
    eps_values: Sequence[float],
    n_leapfrog: int,
    mass: Mass = None,
    thin: int = 1,
    want_traj: bool = False,
    want_diag: bool = False,
    want_margins: bool = False,
    forced_accept: Optional[torch.Tensor] = None,
):
    """T = len(eps_values) transitions; ``p_noise[t]`` are the standard normals of transition
    t (scaled by sqrt(mass) here, hmc.py:118-133), ``u[t]`` the accept uniforms.

    Returns a dict: ``x`` final state, ``accepted`` bool [T, n], ``margin`` = min |u - a| over
    all decisions (how far the closest accept/reject call was from flipping; ``margins`` [T, n]: per decision), optional
    ``trajectory`` [n, T // thin, dim] and ``diagnostics`` (mean, var, energy, acceptance_rate).
    ``forced_accept`` (bool [T, n]): take these decisions instead of ``u < a`` -- the fp64 referee run follows the
    fp32 run's accept path (tests/golden/make_referee.py).
    """
    x = x0.clone()
    n, dim = x.shape
    T = len(eps_values)
    n_kept = T // thin
    accepted_all = torch.empty(T, n, dtype=torch.bool)
    margins_all = torch.full((T, n), float("inf"), dtype=torch.float64) if want_margins else None  # |u - a| of every decision
    margin = float("inf")
    traj = torch.empty(n, n_kept, dim, dtype=x.dtype) if want_traj else None
    diag: Optional[Dict[str, torch.Tensor]] = None
    if want_diag:
        diag = {
            "mean": torch.empty(n_kept, dim, dtype=x.dtype),
            "var": torch.empty(n_kept, dim, dtype=x.dtype),
            "energy": torch.empty(n_kept, dtype=x.dtype),
            "acceptance_rate": torch.empty(n_kept, dtype=x.dtype),
        }
    keep = 0
    for t in range(T):
        p = p_noise[t].clone()
        if mass is not None:
            if isinstance(mass, float):
                p.mul_(math.sqrt(mass))
            else:
                p.mul_(torch.sqrt(mass).view((1,) * (p.ndim - 1) + (-1,)))
        h0 = energy.energy(x).clamp_(min=-1e10, max=1e10) + kinetic(p, mass).clamp_(min=0.0, max=1e10)
        xp, pp = leapfrog(energy, x, p, eps_values[t], n_leapfrog, mass, safe=True)
        h1 = energy.energy(xp).clamp_(min=-1e10, max=1e10) + kinetic(pp, mass).clamp_(min=0.0, max=1e10)
        delta = (h0 - h1).clamp_(min=-50.0, max=50.0)
        a = torch.exp(delta).clamp_(max=1.0)
        acc = u[t] < a
        if forced_accept is not None:
            acc = forced_accept[t].clone()
        finite = torch.isfinite(a)
        if bool(finite.any()):
            margin = min(margin, float((u[t][finite] - a[finite]).abs().min()))
        accepted_all[t] = acc
        if margins_all is not None:
            margins_all[t] = torch.where(finite, (u[t] - a).abs().double(), torch.zeros((), dtype=torch.float64))
        x = torch.where(acc.view(-1, 1), xp, x)
        if (t + 1) % thin == 0:
            if traj is not None:
                traj[:, keep, :] = x
            if diag is not None:
                diag["mean"][keep] = x.mean(dim=0)
                diag["var"][keep] = (
                    x.var(dim=0, unbiased=False).clamp_(min=1e-10, max=1e10) if n > 1 else torch.zeros(dim, dtype=x.dtype)
                )
                diag["energy"][keep] = energy.energy(x).clamp_(min=-1e10, max=1e10).mean()
                diag["acceptance_rate"][keep] = acc.float().mean()
            keep += 1
    return {"x": x, "accepted": accepted_all, "margin": margin, "margins": margins_all, "trajectory": traj, "diagnostics": diag}
