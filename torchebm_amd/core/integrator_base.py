"""Integrator base classes: the plug-in socket of the samplers.

Host-side mirror of the reference's torchebm/core/base_integrator.py for the part of it
that is on the Langevin/HMC path:

* ``BaseIntegrator``                 (:11-92)   abstract ``step`` / ``integrate``
* ``BaseSDERungeKuttaIntegrator``    (:627-817) explicit tableau + additive Wiener noise
* ``BaseSymplecticIntegrator``       (:820-889) phase-space state, safe-mode helpers

Dispatch rule used by every concrete integrator here: a CUDA fp32 state goes through the
HIP library (``_lib``), which raises if it is missing; a CPU state is advanced with eager
torch ops in the reference's operation order (BASELINE config 1 is that CPU path).
Adaptive / implicit Runge-Kutta machinery (:95-624) is out of scope.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Callable, Dict, Optional, Sequence, Tuple

import torch

from .. import _lib, _rng
from .module import TorchEBMModule, warn_once

Drift = Callable[[torch.Tensor, torch.Tensor], torch.Tensor]


def on_hip_path(x: torch.Tensor) -> bool:
    """True when ``x`` must be processed by the HIP kernels (CUDA device, fp32)."""
    return x.is_cuda and x.dtype == torch.float32


class BaseIntegrator(TorchEBMModule, ABC):
    """Advances a state dict (``{"x"}`` or ``{"x", "p"}``) under caller-supplied dynamics."""

    def __init__(self, device: Optional[torch.device] = None, dtype: Optional[torch.dtype] = None, *args, **kwargs):
        super().__init__(device=device, dtype=dtype, *args, **kwargs)

    @staticmethod
    def _resolve_drift(drift: Optional[Drift]) -> Drift:
        if drift is None:
            raise ValueError(
                "drift must be provided explicitly. For EBM sampling, pass "
                "drift=lambda x, t: -model.gradient(x) from the caller."
            )
        return drift

    @abstractmethod
    def step(self, state: Dict[str, torch.Tensor], step_size, *args, **kwargs) -> Dict[str, torch.Tensor]:
        """One integrator application; returns a new state dict with the same keys."""

    @abstractmethod
    def integrate(self, state: Dict[str, torch.Tensor], step_size, n_steps: int, *args, **kwargs) -> Dict[str, torch.Tensor]:
        """``n_steps`` integrator applications."""


class BaseSDERungeKuttaIntegrator(BaseIntegrator):
    r"""Explicit Runge-Kutta drift update followed by an Euler-order Wiener increment:

    .. math:: x_{n+1} = x_n + h \sum_i b_i k_i + \sqrt{2 D}\,\Delta W_n .

    Subclasses provide ``tableau_a`` (strictly lower-triangular rows), ``tableau_b`` and
    ``tableau_c``.  The arithmetic follows base_integrator.py:300-347,387-397,711-731.
    """

    @property
    @abstractmethod
    def tableau_a(self) -> Sequence[Sequence[float]]: ...

    @property
    @abstractmethod
    def tableau_b(self) -> Sequence[float]: ...

    @property
    @abstractmethod
    def tableau_c(self) -> Sequence[float]: ...

    def _is_forward_euler(self) -> bool:
        return tuple(self.tableau_b) == (1.0,) and all(len(r) == 0 for r in self.tableau_a)

    # ---- eager torch arithmetic (CPU states; reference op order) ---------------------
    def _drift_update(self, x: torch.Tensor, h, drift_fn: Drift, t: torch.Tensor) -> torch.Tensor:
        a, b, c = self.tableau_a, self.tableau_b, self.tableau_c
        stages = x.new_empty((len(b),) + tuple(x.shape))
        for i in range(len(b)):
            xi = x
            if i > 0:
                coeff = torch.tensor(list(a[i][:i]), dtype=x.dtype, device=x.device)
                xi = x + h * torch.einsum("i,i...->...", coeff, stages[:i])
            stages[i] = drift_fn(xi, t + c[i] * h)
        weights = torch.tensor(list(b), dtype=x.dtype, device=x.device)
        return x + h * torch.einsum("i,i...->...", weights, stages)

    def step(
        self,
        state: Dict[str, torch.Tensor],
        step_size,
        *,
        drift: Optional[Drift] = None,
        diffusion: Optional[torch.Tensor] = None,
        noise: Optional[torch.Tensor] = None,
        noise_scale=None,
        t: Optional[torch.Tensor] = None,
        generator: Optional[torch.Generator] = None,
    ) -> Dict[str, torch.Tensor]:
        """One step.  ``noise`` bypasses the RNG; ``diffusion`` (tensor ``D``) is the
        alternative to the scalar ``noise_scale`` (``D = noise_scale**2``)."""
        x = state["x"]
        drift_fn = self._resolve_drift(drift)
        if t is None:
            t = torch.zeros(x.size(0), device=x.device, dtype=x.dtype)

        scalar_noise = diffusion is None  # D is a Python float (or absent)
        if on_hip_path(x) and self._is_forward_euler() and not torch.is_tensor(step_size):
            if scalar_noise and not torch.is_tensor(noise_scale):
                return {"x": self._hip_em_step(x, float(step_size), drift_fn(x, t), noise, noise_scale, generator)}
            d = diffusion if diffusion is not None else noise_scale**2  # (a tensor noise_scale: D = noise_scale ** 2)
            if torch.is_tensor(d) and d.is_cuda and d.device == x.device and d.dtype == torch.float32:
                return {"x": self._hip_em_step_diffusion(x, float(step_size), drift_fn(x, t), noise, d, generator)}

        if x.is_cuda:
            warn_once(
                "sde-rk-eager-cuda",
                "torchebm_amd: this SDE step configuration (non-fp32 state, tensor diffusion or a multi-stage "
                "tableau) is not accelerated by the HIP kernels; running eager torch ops on the GPU.",
                UserWarning,
            )
        x_new = self._drift_update(x, step_size, drift_fn, t)
        d_val = diffusion if diffusion is not None else (None if noise_scale is None else noise_scale**2)
        if d_val is not None:
            if noise is None:
                noise = torch.randn_like(x, device=self.device, dtype=self.dtype, generator=generator)
            dw = noise * (step_size**0.5)
            x_new = x_new + (2.0 * d_val) ** 0.5 * dw
        return {"x": x_new}

    def _hip_em_step_diffusion(self, x, eta: float, drift_val, noise, d: torch.Tensor, generator) -> torch.Tensor:
        """Tensor diffusion coefficient (base_integrator.py:652-671): ``x + h k0 + (2 D) ** 0.5 * (eps * h ** 0.5)`` in one
        launch of ``ebm_langevin_step_diffusion_f32``; D is a 0-dim tensor, one value per trailing coordinate, or
        anything else broadcastable to ``x`` (then materialised as a full field)."""
        xin = _lib.dense_f32(x)
        dv = _lib.dense_f32(drift_val)
        out = torch.empty_like(xin)
        if d.numel() == 1:
            dd, period = d.reshape(1), 1
        elif d.numel() == x.shape[-1] and d.shape[-1] == x.shape[-1]:
            dd, period = d.reshape(-1), x.shape[-1]
        else:
            dd, period = torch.broadcast_to(d, x.shape), xin.numel()
        dd = _lib.dense_f32(dd)
        seed, step = (0, 0) if noise is not None else _rng.reserve(generator, x.device, 1)
        nz = None if noise is None else _lib.dense_f32(noise.to(x.device))
        _lib.call(
            "ebm_langevin_step_diffusion_f32",
            _lib.ptr(xin), _lib.ptr(dv), _lib.ptr(out), _lib.ptr(nz), _lib.ptr(dd), period, xin.numel(),
            -eta, eta**0.5, seed, step, _lib.stream_handle(x.device),
        )
        return out.view_as(x)

    def _hip_em_step(self, x, eta: float, drift_val, noise, noise_scale, generator) -> torch.Tensor:
        xin = _lib.dense_f32(x)
        dv = _lib.dense_f32(drift_val)
        out = torch.empty_like(xin)
        if noise_scale is None:
            coef, seed, step = 0.0, 0, 0
        else:
            coef = (2.0 * float(noise_scale) ** 2) ** 0.5
            seed, step = (0, 0) if noise is not None else _rng.reserve(generator, x.device, 1)
        nz = None if (noise is None or noise_scale is None) else _lib.dense_f32(noise.to(x.device))
        # the kernel computes x - eta*grad; with grad := drift and eta := -eta that is
        # x + fl(eta*drift), bit for bit the reference's x + h*(1.0*k0)
        _lib.call(
            "ebm_langevin_step_f32",
            _lib.ptr(xin), _lib.ptr(dv), _lib.ptr(out), _lib.ptr(nz), xin.numel(),
            -eta, eta**0.5, coef, 0, 0.0, 0.0, seed, step, _lib.stream_handle(x.device),
        )
        return out.view_as(x)

    def integrate(
        self,
        state: Dict[str, torch.Tensor],
        step_size,
        n_steps: int,
        *,
        drift: Optional[Drift] = None,
        diffusion: Optional[Callable[[torch.Tensor, torch.Tensor], torch.Tensor]] = None,
        noise_scale=None,
        t: Optional[torch.Tensor] = None,
        adaptive: Optional[bool] = None,
        inference_mode: bool = False,
        generator: Optional[torch.Generator] = None,
    ) -> Dict[str, torch.Tensor]:
        """Fixed-step SDE/ODE integration over ``n_steps`` (base_integrator.py:733-817).
        Adaptive stepping belongs to the ODE-flow machinery and is not provided."""
        if inference_mode:
            with torch.inference_mode():
                return self.integrate(
                    state, step_size, n_steps, drift=drift, diffusion=diffusion, noise_scale=noise_scale,
                    t=t, adaptive=adaptive, generator=generator,
                )
        if adaptive:
            raise NotImplementedError("adaptive Runge-Kutta stepping is outside the Langevin/HMC hot path")
        if n_steps <= 0:  # base_integrator.py:418-419 (checked before the grid, whether or not one is passed)
            raise ValueError("n_steps must be positive")
        x = state["x"]
        drift_fn = self._resolve_drift(drift)
        if t is None:
            h = step_size if torch.is_tensor(step_size) else torch.tensor(step_size, dtype=x.dtype, device=x.device)
            t = torch.arange(n_steps + 1, dtype=x.dtype, device=x.device) * h
        elif t.ndim != 1 or t.numel() < 2:
            raise ValueError("t must be a 1D tensor with length >= 2")
        d_const = None if (diffusion is not None or noise_scale is None) else noise_scale**2
        batch = x.size(0)
        for i in range(t.numel() - 1):
            dt = t[i + 1] - t[i]
            tb = t[i].expand(batch)
            d_val = diffusion(x, tb) if diffusion is not None else d_const
            x = self._drift_update(x, dt, drift_fn, tb)
            if d_val is not None:
                eps = torch.randn_like(x, generator=generator)
                x = x + (2.0 * d_val) ** 0.5 * eps * torch.sqrt(dt)  # note: different op order from step()
        return {"x": x}


class BaseSymplecticIntegrator(BaseIntegrator):
    """Phase-space integrators for HMC.  ``separable = True`` means ``H = U(x) + K(p)``
    with a ``drift(x, t) = -grad U`` callback and an optional mass (base_integrator.py:820-889)."""

    separable: bool = True
    _SAFE_CLAMP: float = 1e6

    @staticmethod
    def _validate_n_steps(n_steps: int) -> None:
        if n_steps <= 0:
            raise ValueError("n_steps must be positive")

    @staticmethod
    def _unpack_state(state: Dict[str, torch.Tensor], step_size) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        x, p = state["x"], state["p"]
        if not torch.is_tensor(step_size):
            # a fill kernel, not a host-to-device copy: legal inside HIP-graph capture, same fp32 value
            step_size = torch.full((), float(step_size), device=x.device, dtype=x.dtype)
        t = torch.zeros(x.size(0), device=x.device, dtype=x.dtype)
        return x, p, step_size, t

    def _safe_clamp_(self, tensor: torch.Tensor) -> torch.Tensor:
        return tensor.clamp_(min=-self._SAFE_CLAMP, max=self._SAFE_CLAMP)

    @staticmethod
    def _sanitize_state_(x: torch.Tensor, p: torch.Tensor) -> None:
        # unconditional: a data-dependent isnan().any() would be a host sync
        x.nan_to_num_(nan=0.0)
        p.nan_to_num_(nan=0.0)
