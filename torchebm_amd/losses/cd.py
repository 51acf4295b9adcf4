"""Contrastive divergence loss (reference: torchebm/losses/contrastive_divergence.py:15-223).

Config 5's caller of the sampler: negatives come from ``sampler.sample`` started at the
data (CD-k) or at the persistent replay buffer (PCD).  The sampler runs under ``no_grad``
and returns a detached tensor, so nothing here differentiates through the chain.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch

from ..core.loss_base import BaseContrastiveDivergence
from ..core.module import warn_once


class ContrastiveDivergence(BaseContrastiveDivergence):
    r"""``L = E_data[E(x)] - E_model[E(x^-)] + \lambda (E[E(x)^2] + E[E(x^-)^2])``."""

    _CD_OPTION_KEYS = ("energy_reg_weight", "add_noise_to_real", "noise_scale")

    def __init__(
        self,
        model,
        sampler,
        k_steps=10,
        persistent=False,
        buffer_size=10000,
        init_steps=100,
        new_sample_ratio=0.05,
        energy_reg_weight=0.001,
        add_noise_to_real=False,
        noise_scale=0.0001,
        dtype=torch.float32,
        device=torch.device("cpu"),
        *args,
        **kwargs,
    ):
        super().__init__(
            *args, model=model, sampler=sampler, k_steps=k_steps, persistent=persistent, buffer_size=buffer_size,
            new_sample_ratio=new_sample_ratio, init_steps=init_steps, dtype=dtype, device=device, **kwargs,
        )
        self.energy_reg_weight = energy_reg_weight
        self.add_noise_to_real = add_noise_to_real
        self.noise_scale = noise_scale

    def forward(
        self,
        x: torch.Tensor,
        *args,
        model_kwargs: Optional[dict] = None,
        generator: Optional[torch.Generator] = None,
        **kwargs,
    ) -> Tuple[torch.Tensor, torch.Tensor]:
        """Returns ``(loss, negative_samples)``."""
        model_kwargs = self._prepare_model_kwargs(model_kwargs)
        if any(k in kwargs for k in self._CD_OPTION_KEYS):
            warn_once(
                "cd-option-kwargs",
                "Passing energy_reg_weight/add_noise_to_real/noise_scale to ContrastiveDivergence.__call__ is "
                "deprecated; set them on the constructor instead.",
            )
        starts = self.get_start_points(x, generator=generator)
        negatives = self.sampler.sample(x=starts, n_steps=self.k_steps, model_kwargs=model_kwargs, generator=generator)
        if self.persistent:
            with torch.no_grad():
                self.update_buffer(negatives)
        for key in self._CD_OPTION_KEYS:
            kwargs.setdefault(key, getattr(self, key))
        loss = self.compute_loss(x, negatives, *args, model_kwargs=model_kwargs, generator=generator, **kwargs)
        return loss, negatives

    def compute_loss(
        self,
        x: torch.Tensor,
        pred_x: torch.Tensor,
        *args,
        model_kwargs: Optional[dict] = None,
        generator: Optional[torch.Generator] = None,
        **kwargs,
    ) -> torch.Tensor:
        x = x.to(self.device, self.dtype)
        pred_x = pred_x.to(self.device, self.dtype)
        cond = model_kwargs or {}
        with torch.set_grad_enabled(True):
            if kwargs.get("add_noise_to_real", self.add_noise_to_real):
                jitter = kwargs.get("noise_scale", self.noise_scale) * torch.randn_like(x, generator=generator)
                real = x + jitter
            else:
                real = x
            if (getattr(self.model, "ROWS_INDEPENDENT", False) and not cond and real.is_cuda and real.shape == pred_x.shape
                    and not real.requires_grad and not pred_x.requires_grad):
                # one evaluation of both halves for an energy that declares every row's value independent of the batch around it
                # (MLPEnergy): the same arithmetic per row in half the launches -- the loss's forward / backward is some sixty
                # small kernels whose dispatch gaps, not their work, are a tenth of a captured training step
                e_both = self.model(torch.cat((real, pred_x)))
                # both halves' statistics from the [2, n] view: two row reductions instead of four means and two squares (every one
                # of these is a 10 us graph node on a 65 536-element vector)
                e2 = e_both.view(2, real.shape[0])
                means = e2.mean(dim=1)
                loss = means[0] - means[1]
                reg = kwargs.get("energy_reg_weight", self.energy_reg_weight)
                if reg > 0:
                    loss = loss + reg * e2.square().mean(dim=1).sum()
                return torch.where(torch.isfinite(loss), loss, loss.new_full((), 0.1))
            e_data = self.model(real, **cond)
            e_model = self.model(pred_x, **cond)
        loss = torch.mean(e_data) - torch.mean(e_model)
        reg = kwargs.get("energy_reg_weight", self.energy_reg_weight)
        if reg > 0:
            loss = loss + reg * (torch.mean(e_data**2) + torch.mean(e_model**2))
        # a non-finite loss must not poison the optimiser: constant fallback, no host sync
        # (new_full: a fill kernel -- torch.tensor(0.1, device=...) is a pageable host-to-device copy, which a HIP graph cannot capture)
        return torch.where(torch.isfinite(loss), loss, loss.new_full((), 0.1))
