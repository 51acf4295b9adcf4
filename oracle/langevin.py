"""Oracle: Euler-Maruyama step and the Langevin chain with injected noise.

Follows torchebm/core/base_integrator.py:673-731 (step), :387-397 (stage combine),
torchebm/samplers/langevin_dynamics.py:154-185 (loop, clamp, thinning, diagnostics).
"""

from __future__ import annotations

from typing import Dict, Optional, Sequence, Tuple

import torch


def em_step(x: torch.Tensor, grad: torch.Tensor, eps: Optional[torch.Tensor], eta: float, sigma: Optional[float]):
    """One step.  ``eta``/``sigma`` are Python floats; every line is one rounded tensor op,
    exactly the reference's sequence:
        k0 = drift = -grad                                   langevin_dynamics.py:154
        x1 = x + eta * (1.0 * k0)                            base_integrator.py:397 (b = [1.])
        dw = eps * eta**0.5                                  :728
        x' = x1 + (2.0 * sigma**2)**0.5 * dw                 :729
    """
    k0 = -grad
    x1 = x + eta * (1.0 * k0)
    if sigma is None or eps is None:
        return x1
    dw = eps * (eta**0.5)
    return x1 + (2.0 * sigma**2) ** 0.5 * dw


def heun_step(energy, x: torch.Tensor, eps: Optional[torch.Tensor], eta: float, sigma: Optional[float]):
    """One Heun-SDE step (torchebm/integrators/heun.py: a = ((), (1,)), b = (1/2, 1/2)), in the op order
    of the generic tableau code (base_integrator.py:387-397), then the Euler-order noise (:728-729):
        k0 = -grad(x)
        x1 = x + eta * einsum([1.], [k0])                    stage input
        k1 = -grad(x1)
        xd = x + eta * einsum([.5, .5], [k0, k1])            drift update
        x' = xd + (2 sigma^2)^0.5 * (eps * eta^0.5)
    """
    k0 = -energy.grad(x)
    stages = torch.stack([k0])
    x1 = x + eta * torch.einsum("i,i...->...", torch.tensor([1.0], dtype=x.dtype), stages)
    k1 = -energy.grad(x1)
    stages = torch.stack([k0, k1])
    xd = x + eta * torch.einsum("i,i...->...", torch.tensor([0.5, 0.5], dtype=x.dtype), stages)
    if sigma is None or eps is None:
        return xd
    dw = eps * (eta**0.5)
    return xd + (2.0 * sigma**2) ** 0.5 * dw


def langevin_chain(
    energy,
    x0: torch.Tensor,
    noise: torch.Tensor,
    etas: Sequence[float],
    sigmas: Sequence[float],
    clamp: Optional[Tuple[float, float]] = None,
    thin: int = 1,
    want_traj: bool = False,
    want_diag: bool = False,
    integrator: str = "euler_maruyama",
):
    """k = len(etas) steps with ``noise[i]`` as the step-i Wiener draw; ``integrator`` is
    ``"euler_maruyama"`` (default) or ``"heun"``.

    Returns ``(x_final, trajectory_or_None, diagnostics_or_None)``; trajectory is
    ``[n, k // thin, dim]`` and diagnostics are ``mean``/``var``/``energy`` per kept step
    (biased variance clamped to [1e-10, 1e10]; n == 1 special case) as in
    langevin_dynamics.py:170-185.
    """
    x = x0.clone()
    k = len(etas)
    n = x.shape[0]
    n_kept = k // thin
    traj = torch.empty((n, n_kept) + tuple(x.shape[1:]), dtype=x.dtype) if want_traj else None
    diag: Optional[Dict[str, torch.Tensor]] = None
    if want_diag:
        diag = {
            "mean": torch.empty((n_kept,) + tuple(x.shape[1:]), dtype=x.dtype),
            "var": torch.empty((n_kept,) + tuple(x.shape[1:]), dtype=x.dtype),
            "energy": torch.empty(n_kept, dtype=x.dtype),
        }
    keep = 0
    for i in range(k):
        if integrator == "heun":
            x = heun_step(energy, x, noise[i], etas[i], sigmas[i])
        else:
            x = em_step(x, energy.grad(x), noise[i], etas[i], sigmas[i])
        if clamp is not None:
            x = x.clamp_(*clamp)
        if (i + 1) % thin == 0:
            if traj is not None:
                traj[:, keep] = x
            if diag is not None:
                if n > 1:
                    diag["mean"][keep] = x.mean(dim=0)
                    diag["var"][keep] = x.var(dim=0, unbiased=False).clamp_(min=1e-10, max=1e10)
                else:
                    diag["mean"][keep] = x.squeeze(0)
                    diag["var"][keep].zero_()
                diag["energy"][keep] = energy.energy(x).mean()
            keep += 1
    return x, traj, diag
