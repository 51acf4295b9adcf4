"""Noise-free descent samplers: gradient descent and Nesterov momentum.

Reference: torchebm/samplers/gradient_descent.py (GradientDescentSampler :15-140, NesterovSampler
:143-281).  Same ``sample`` contract as the MCMC samplers; the only diagnostic is ``"energy"``.
These are the ``noise = 0`` relatives of the Langevin chain: on a CUDA fp32
``[n, dim]`` state with an analytic energy all k steps run in one ``ebm_descent_chain_f32`` launch,
otherwise each step is ``model.gradient`` + one fused update launch.

Arithmetic note: ``torch.sub(x, g, alpha=eta)`` and ``torch.add(x, v, alpha=mu)`` are single-rounding
FMAs in the reference's CPU kernels; the HIP kernels use FMAs in the same places.
"""

from __future__ import annotations

from typing import Any, Dict, Optional, Tuple, Union

import torch

from .. import _lib
from ..core.energies import BaseModel, fused_spec_for
from ..core.sampler_base import BaseSampler
from ..core.schedules import BaseScheduler


class _DescentBase(BaseSampler):
    _nesterov = False
    momentum = 0.0

    def _route(self, x: torch.Tensor, model_kwargs: Dict[str, Any]):
        if not x.is_cuda or x.dtype != torch.float32 or (self.use_mixed_precision and self.autocast_available):
            return "eager", None
        spec = fused_spec_for(self.model, x, model_kwargs)
        if spec is not None and spec.langevin_only:
            spec = None
        return ("fused", spec) if spec is not None else ("step", None)

    @torch.no_grad()
    def sample(
        self,
        x: Optional[torch.Tensor] = None,
        dim: Optional[Union[int, Tuple[int, ...]]] = None,
        n_steps: int = 100,
        n_samples: int = 1,
        thin: int = 1,
        return_trajectory: bool = False,
        return_diagnostics: bool = False,
        reset_schedulers: bool = True,
        *,
        model_kwargs: Optional[Dict[str, Any]] = None,
        generator: Optional[torch.Generator] = None,
    ) -> Union[torch.Tensor, Tuple[torch.Tensor, Dict[str, torch.Tensor]]]:
        """Run ``n_steps`` descent steps from ``x`` (or from N(0, I) draws; ``generator`` is only
        used for that initial draw).  Diagnostics: ``"energy"`` ``[n_steps // thin]``."""
        if thin < 1:
            raise ValueError("thin must be >= 1")
        if reset_schedulers:
            self.reset_schedulers()
        x = self._init_state(x, dim, n_samples, generator)
        model_kwargs = self._prepare_model_kwargs(model_kwargs)
        route, spec = self._route(x, model_kwargs)
        n_kept = n_steps // thin
        traj = (
            torch.empty(x.shape[0], n_kept, *x.shape[1:], device=x.device, dtype=x.dtype) if return_trajectory else None
        )
        diag = {"energy": torch.empty(n_kept, dtype=self.dtype, device=self.device)} if return_diagnostics else None
        if route == "fused":
            out = self._run_fused(x, spec, n_steps, thin, traj, diag)
        else:
            out = self._run_stepwise(x, model_kwargs, n_steps, thin, traj, diag, hip=(route == "step"))
        result = traj if return_trajectory else out
        return (result, diag) if return_diagnostics else result

    # ---- per-step loop -------------------------------------------------------------------
    def _run_stepwise(self, x, model_kwargs, n_steps, thin, traj, diag, hip: bool):
        mu = self.momentum
        v = torch.zeros_like(x) if self._nesterov else None
        if hip:
            x = _lib.dense_f32(x)
            stream = _lib.stream_handle(x.device)
            if v is not None:
                v = torch.zeros_like(x)
        keep = 0
        with self.autocast_context():
            for i in range(n_steps):
                eta = self.get_scheduled_value("step_size")
                if hip:
                    point = x
                    if v is not None:
                        point = torch.empty_like(x)
                        _lib.call("ebm_lookahead_f32", _lib.ptr(x), _lib.ptr(v), _lib.ptr(point), x.numel(), mu, stream)
                    grad = _lib.dense_f32(self._model_gradient(point, model_kwargs))
                    out = torch.empty_like(x)
                    _lib.call("ebm_descent_step_f32", _lib.ptr(x), _lib.ptr(grad), _lib.ptr(v), _lib.ptr(out), x.numel(),
                              eta, mu, stream)
                    x = out
                elif v is not None:
                    lookahead = torch.add(x, v, alpha=mu)
                    grad = self._model_gradient(lookahead, model_kwargs)
                    v.mul_(mu).sub_(grad, alpha=eta)
                    x = x + v
                else:
                    x = torch.sub(x, self._model_gradient(x, model_kwargs), alpha=eta)
                if (i + 1) % thin == 0:
                    if traj is not None:
                        traj[:, keep] = x
                    if diag is not None:
                        diag["energy"][keep] = self._model_energy(x, model_kwargs).mean()
                    keep += 1
                self.step_schedulers()
        return x

    # ---- k-fused kernel --------------------------------------------------------------------
    def _run_fused(self, x, spec, n_steps, thin, traj, diag):
        n, dim = x.shape
        state = _lib.dense_f32(x).clone()
        sched = self.schedulers["step_size"]
        etas = [sched.get_value()] if sched.is_constant() else sched.preview(n_steps)
        stream = _lib.stream_handle(x.device)
        spec_c = spec.to_c()

        def launch(t0, k, thin_, traj_):
            if len(etas) == 1:
                eta, table = etas[0], None
            else:
                eta = etas[t0]
                table = torch.tensor(etas[t0 : t0 + k], dtype=torch.float32).to(x.device, non_blocking=True)
            _lib.call("ebm_descent_chain_f32", spec_c, _lib.ptr(state), n, dim, k, eta, _lib.ptr(table),
                      int(self._nesterov), self.momentum, thin_, _lib.ptr(traj_), stream)

        if n_steps > 0 and n > 0:
            if diag is None or self._nesterov:
                # (Nesterov's velocity lives inside the launch, so the whole run must be one launch;
                #  its per-kept-step energies are then read off the stored trajectory)
                need_traj = traj if (traj is not None or diag is None) else torch.empty(n, n_steps // thin, dim, device=x.device)
                launch(0, n_steps, thin, need_traj)
                if diag is not None:
                    energy = torch.empty(n, dtype=torch.float32, device=x.device)
                    for keep in range(n_steps // thin):
                        rows = need_traj[:, keep].contiguous()
                        _lib.call("ebm_energy_grad_f32", spec_c, _lib.ptr(rows), n, dim, _lib.ptr(energy), None, stream)
                        diag["energy"][keep] = energy.mean()
            else:
                energy = torch.empty(n, dtype=torch.float32, device=x.device)
                done = 0
                for keep in range(n_steps // thin):
                    launch(done, thin, thin, None)
                    done += thin
                    if traj is not None:
                        traj[:, keep] = state
                    _lib.call("ebm_energy_grad_f32", spec_c, _lib.ptr(state), n, dim, _lib.ptr(energy), None, stream)
                    diag["energy"][keep] = energy.mean()
                if done < n_steps:
                    launch(done, n_steps - done, thin, None)
        self.advance_schedulers(n_steps)
        return state


class GradientDescentSampler(_DescentBase):
    r"""``x \leftarrow x - \eta \nabla E(x)``."""

    def __init__(
        self,
        model: BaseModel,
        step_size: Union[float, BaseScheduler] = 1e-3,
        dtype: torch.dtype = torch.float32,
        device: Optional[Union[str, torch.device]] = None,
    ):
        super().__init__(model=model, dtype=dtype, device=device)
        self._register_param("step_size", step_size, positive=True)


class NesterovSampler(_DescentBase):
    r"""Nesterov accelerated descent: gradient at the look-ahead point ``x + \mu v``."""

    _nesterov = True

    def __init__(
        self,
        model: BaseModel,
        step_size: Union[float, BaseScheduler] = 1e-3,
        momentum: float = 0.9,
        dtype: torch.dtype = torch.float32,
        device: Optional[Union[str, torch.device]] = None,
    ):
        super().__init__(model=model, dtype=dtype, device=device)
        if not 0 <= momentum < 1:
            raise ValueError("momentum must be in [0, 1)")
        self.momentum = momentum
        self._register_param("step_size", step_size, positive=True)
