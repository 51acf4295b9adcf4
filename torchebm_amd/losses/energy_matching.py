"""The contrastive branch of Energy Matching (reference: torchebm/losses/energy_matching.py:318-372
for the negatives, :456-478 for the term) -- SURVEY.md §8(f) n2, the second in-tree caller of the
Langevin path.

Per training step two Langevin calls produce the negatives:

* ``round(B * noise_fraction)`` chains start at source samples (N(0, I) by default) and sweep the
  transport-to-Boltzmann temperature profile ``TemperatureScheduler(epsilon_max, tau_star, n)``:
  sigma_i = 0 while t_i < tau_star (pure gradient flow), then sqrt(eps(t_i)) up to sqrt(eps_max);
* the rest start at randomly chosen data rows and are held at ``sqrt(epsilon_max)``.

On an analytic / packaged-MLP energy both calls are ONE fused launch each: the sweep becomes the
kernel's per-step coefficient table (``float[k][4]``), which is what this row exercises.  The
flow-matching half of the reference loss (OT coupling, interpolant, second-order backward) is a
regression objective outside the sampler path and is not part of this package.
"""

from __future__ import annotations

import math
from typing import Any, Dict, Optional

import torch

from ..core.loss_base import BaseLoss
from ..core.schedules import ConstantScheduler, TemperatureScheduler
from ..samplers.langevin import LangevinDynamics


def trimmed_mean(values: torch.Tensor, trim_fraction: float) -> torch.Tensor:
    """Mean of ``values`` without its ``int(trim_fraction * n)`` largest entries
    (reference: losses/loss_utils.py:20-44)."""
    if not 0.0 <= trim_fraction < 1.0:
        raise ValueError(f"trim_fraction must be in [0, 1), got {trim_fraction}")
    drop = int(trim_fraction * values.shape[0])
    if drop == 0:
        return values.mean()
    return values.sort().values[: values.shape[0] - drop].mean()


class EnergyMatchingContrastive(BaseLoss):
    r"""``lambda_cd * (E_data[V(x)] - trimmed_mean(V(x^-)))`` floored at ``-cd_clamp``, with the
    negatives ``x^-`` from the two temperature-scheduled Langevin calls described above.

    Constructor arguments carry the names, defaults and validation of the reference's
    ``EnergyMatchingLoss`` for this branch (energy_matching.py:144-199).  The loss owns the sampler's
    ``noise_scale`` scheduler while it runs, exactly as the reference does.
    """

    def __init__(
        self,
        model,
        sampler: Optional[LangevinDynamics] = None,
        lambda_cd: float = 2.0,
        epsilon_max: float = 0.15,
        tau_star: float = 0.8,
        n_langevin_steps: int = 200,
        langevin_dt: float = 0.01,
        noise_fraction: float = 0.5,
        cd_trim_fraction: float = 0.1,
        cd_clamp: Optional[float] = 0.02,
        dtype: torch.dtype = torch.float32,
        device=None,
    ):
        super().__init__(dtype=dtype, device=device)
        if not 0.0 <= noise_fraction <= 1.0:
            raise ValueError(f"noise_fraction must be in [0, 1], got {noise_fraction}")
        if not 0.0 <= cd_trim_fraction < 1.0:
            raise ValueError(f"cd_trim_fraction must be in [0, 1), got {cd_trim_fraction}")
        if cd_clamp is not None and cd_clamp < 0:
            raise ValueError(f"cd_clamp must be >= 0 or None, got {cd_clamp}")
        if langevin_dt <= 0:
            raise ValueError(f"langevin_dt must be positive, got {langevin_dt}")
        self.model = model
        self.sampler = sampler if sampler is not None else LangevinDynamics(
            model=model, step_size=langevin_dt, noise_scale=1.0, dtype=dtype, device=device
        )
        self._register_param("lambda_cd", lambda_cd)
        self.epsilon_max = epsilon_max
        self.tau_star = tau_star
        self.n_langevin_steps = n_langevin_steps
        self.langevin_dt = langevin_dt
        self.noise_fraction = noise_fraction
        self.cd_trim_fraction = cd_trim_fraction
        self.cd_clamp = cd_clamp
        # sample() resets registered schedulers on entry, so one instance of each serves every step
        self._noise_sweep = TemperatureScheduler(epsilon_max=epsilon_max, tau_star=tau_star, n_steps=n_langevin_steps)
        self._noise_const = ConstantScheduler(math.sqrt(epsilon_max))

    @property
    def lambda_cd(self) -> float:
        return self.get_scheduled_value("lambda_cd")

    @lambda_cd.setter
    def lambda_cd(self, value) -> None:
        self._register_param("lambda_cd", value)

    # ------------------------------------------------------------------------------------
    def sample_negatives(
        self,
        x1: torch.Tensor,
        x0: Optional[torch.Tensor] = None,
        model_kwargs: Optional[Dict[str, Any]] = None,
        generator: Optional[torch.Generator] = None,
    ) -> torch.Tensor:
        """Detached negatives ``[B, ...]``: source-initialised sweep chains first, data-initialised
        constant-temperature chains after them.  Draw order on ``generator`` (randn / randperm, chain
        noise, randperm, chain noise) follows energy_matching.py:318-372.  Batch-aligned conditioning
        tensors are sliced per part; the conditioning aligned with the returned rows is left in
        ``self._neg_model_kwargs``."""
        cond = model_kwargs or {}
        batch = x1.shape[0]
        n_noise = int(round(batch * self.noise_fraction))

        def rows(index):
            return {k: (v[index] if torch.is_tensor(v) and v.shape[0] == batch else v) for k, v in cond.items()}

        parts, cond_parts = [], []
        if n_noise > 0:
            part_cond = rows(slice(0, n_noise))
            if x0 is None:
                start = torch.randn(x1[:n_noise].shape, dtype=x1.dtype, device=x1.device, generator=generator)
            else:
                start = x0[torch.randperm(x0.shape[0], device=x0.device, generator=generator)[:n_noise]]
            self.sampler.register_scheduler("noise_scale", self._noise_sweep)
            parts.append(self.sampler.sample(
                x=start.detach(), n_steps=self.n_langevin_steps, model_kwargs=part_cond, generator=generator))
            cond_parts.append(part_cond)
        if batch - n_noise > 0:
            pick = torch.randperm(batch, device=x1.device, generator=generator)[: batch - n_noise]
            part_cond = rows(pick)
            self.sampler.register_scheduler("noise_scale", self._noise_const)
            parts.append(self.sampler.sample(
                x=x1[pick].detach(), n_steps=self.n_langevin_steps, model_kwargs=part_cond, generator=generator))
            cond_parts.append(part_cond)
        self._neg_model_kwargs = {
            k: (torch.cat([p[k] for p in cond_parts], dim=0) if torch.is_tensor(v) and v.shape[0] == batch else v)
            for k, v in cond.items()
        }
        return torch.cat(parts, dim=0).detach()

    def forward(
        self,
        x: torch.Tensor,
        *args,
        x0: Optional[torch.Tensor] = None,
        model_kwargs: Optional[Dict[str, Any]] = None,
        generator: Optional[torch.Generator] = None,
        **kwargs,
    ) -> Dict[str, torch.Tensor]:
        """``{"cd_loss", "cd_value", "negatives"}``; with ``lambda_cd == 0`` the chains are skipped and
        ``cd_loss`` is a zero scalar (the reference's warm-up phase)."""
        cond = self._resolve_model_kwargs(model_kwargs, kwargs, warn_key="em-bare-model-kwargs")
        x = x.to(device=self.device, dtype=self.dtype)
        weight = self.lambda_cd
        if not weight > 0:
            return {"cd_loss": x.new_zeros(())}
        self._neg_model_kwargs = cond
        negatives = self.sample_negatives(x, x0=x0, model_kwargs=cond, generator=generator)
        with self.autocast_context():
            e_pos = self.model(x, **cond)
            e_neg = self.model(negatives, **self._neg_model_kwargs)
        cd_value = e_pos.mean() - trimmed_mean(e_neg, self.cd_trim_fraction)
        cd_loss = weight * cd_value
        if self.cd_clamp is not None:
            cd_loss = torch.clamp(cd_loss, min=-self.cd_clamp)
        return {"cd_loss": cd_loss, "cd_value": cd_value, "negatives": negatives}
