"""Leapfrog (Stormer-Verlet) integrator (reference: torchebm/integrators/leapfrog.py:10-187).

With an opaque ``drift`` closure the force has to come from the caller, so on a CUDA fp32
state each leapfrog step is two fused element-wise launches around the two force
evaluations (``ebm_leapfrog_kick_drift_f32`` / ``ebm_leapfrog_kick_f32``).  When the energy
is one of the analytic models, ``HamiltonianMonteCarlo`` bypasses this class altogether
and runs whole transitions inside ``ebm_hmc_chain_f32``.
"""

from __future__ import annotations

from typing import Dict, Optional, Union

import torch

from .. import _lib
from ..core.integrator_base import BaseSymplecticIntegrator, Drift, on_hip_path
from ..core.module import warn_once

Mass = Optional[Union[float, torch.Tensor]]


def _mass_args(mass: Mass, x: torch.Tensor):
    """(kind, scalar, tensor-or-None) in the form the C ABI takes."""
    if mass is None:
        return _lib.MASS_NONE, 0.0, None
    if isinstance(mass, float):
        return _lib.MASS_SCALAR, mass, None
    diag = _lib.dense_f32(mass.to(x.device).reshape(-1))
    if diag.numel() != x.shape[-1]:
        raise ValueError(f"mass tensor must have {x.shape[-1]} entries, got {diag.numel()}")
    return _lib.MASS_DIAG, 0.0, diag


class LeapfrogIntegrator(BaseSymplecticIntegrator):
    r"""``p_{1/2} = p + \tfrac{\epsilon}{2} f(x)``, ``x' = x + \epsilon\, p_{1/2} / m``,
    ``p' = p_{1/2} + \tfrac{\epsilon}{2} f(x')`` with ``f = -\nabla U``.  ``safe=True`` clamps
    forces to +-1e6 and replaces NaNs in the new state by zeros."""

    separable = True

    def __init__(self, device: Optional[torch.device] = None, dtype: Optional[torch.dtype] = None):
        super().__init__(device=device, dtype=dtype)

    # ---- one step on the HIP path -----------------------------------------------------
    def _hip_step(self, x, p, eps: float, mass: Mass, drift_fn: Drift, t, safe: bool, force=None):
        xin, pin = _lib.dense_f32(x), _lib.dense_f32(p)
        kind, m_scalar, m_diag = _mass_args(mass, xin)
        n_chains = xin.shape[0]
        dim = xin.numel() // max(n_chains, 1)
        stream = _lib.stream_handle(x.device)
        if force is None:  # else: the force the previous step ended on (same position), carried by the caller's opt-in
            force = _lib.dense_f32(drift_fn(x, t))
        x_new, p_half = torch.empty_like(xin), torch.empty_like(pin)
        _lib.call(
            "ebm_leapfrog_kick_drift_f32",
            _lib.ptr(xin), _lib.ptr(pin), _lib.ptr(force), _lib.ptr(x_new), _lib.ptr(p_half),
            n_chains, dim, eps, kind, m_scalar, _lib.ptr(m_diag), int(safe), stream,
        )
        force_new = _lib.dense_f32(drift_fn(x_new.view_as(x), t))
        p_new = torch.empty_like(pin)
        _lib.call(
            "ebm_leapfrog_kick_f32",
            _lib.ptr(x_new), _lib.ptr(p_half), _lib.ptr(force_new), _lib.ptr(p_new),
            xin.numel(), eps, int(safe), stream,
        )
        return x_new.view_as(x), p_new.view_as(p), force_new

    # ---- one step with eager torch ops (CPU states) -------------------------------------
    def _eager_step(self, x, p, eps_t, mass: Mass, drift_fn: Drift, t, safe: bool):
        force = drift_fn(x, t)
        if safe:
            self._safe_clamp_(force)
        p_half = p + 0.5 * eps_t * force
        if mass is None:
            x_new = x + eps_t * p_half
        elif isinstance(mass, float):
            x_new = x + eps_t * p_half / max(mass, 1e-10)
        else:
            shape = (1,) * (x.ndim - 1) + (-1,)
            x_new = x + eps_t * p_half / torch.clamp(mass, min=1e-10).view(shape)
        force_new = drift_fn(x_new, t)
        if safe:
            self._safe_clamp_(force_new)
        p_new = p_half + 0.5 * eps_t * force_new
        if safe:
            self._sanitize_state_(x_new, p_new)
        return x_new, p_new

    def _advance(self, state, step_size, n_steps: int, mass: Mass, drift, safe: bool):
        drift_fn = self._resolve_drift(drift)
        x, p, eps_t, t = self._unpack_state(state, step_size)
        hip = on_hip_path(x) and p.dtype == torch.float32 and not torch.is_tensor(step_size)
        if x.is_cuda and not hip:
            warn_once(
                "leapfrog-eager-cuda",
                "torchebm_amd: leapfrog on a non-fp32 state or with a tensor step size is not accelerated "
                "by the HIP kernels; running eager torch ops on the GPU.",
                UserWarning,
            )
        # ``carry_force`` (an attribute the owning sampler may set, default off): the force at the end of a step is the
        # force at the start of the next -- L + 1 drift evaluations per trajectory instead of the reference's 2 L.  Identical
        # for a deterministic drift on finite states; with ``safe=True`` a position the kick kernel had to scrub keeps
        # the force of the unscrubbed position for one step where the reference re-evaluates.
        carried = None
        for _ in range(n_steps):
            if hip:
                x, p, last = self._hip_step(x, p, float(step_size), mass, drift_fn, t, safe, carried)
                carried = last if getattr(self, "carry_force", False) else None
            else:
                x, p = self._eager_step(x, p, eps_t, mass, drift_fn, t, safe)
        return {"x": x, "p": p}

    def step(
        self,
        state: Dict[str, torch.Tensor],
        step_size=None,
        mass: Mass = None,
        *,
        drift: Optional[Drift] = None,
        safe: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """One leapfrog step; the input tensors are never modified."""
        return self._advance(state, step_size, 1, mass, drift, safe)

    def integrate(
        self,
        state: Dict[str, torch.Tensor],
        step_size=None,
        n_steps: int = None,
        mass: Mass = None,
        *,
        drift: Optional[Drift] = None,
        safe: bool = False,
        inference_mode: bool = False,
    ) -> Dict[str, torch.Tensor]:
        """``n_steps`` leapfrog steps (two force evaluations per step, like the reference)."""
        self._validate_n_steps(n_steps)
        if inference_mode:
            with torch.inference_mode():
                return self._advance(state, step_size, n_steps, mass, drift, safe)
        return self._advance(state, step_size, n_steps, mass, drift, safe)
