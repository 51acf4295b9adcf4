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


class _PairedCDLossHip(torch.autograd.Function):
    """``mean(E+) - mean(E-) + reg (mean(E+^2) + mean(E-^2))`` of the energies ``[E+ | E-]`` of ONE model call, with its analytic
    gradient ``dL/dE_i = (+-1 + 2 reg E_i) / n`` and the reference's guard (a non-finite loss becomes the constant 0.1 and sends
    no gradient: contrastive_divergence.py:150-155).  Two launches of the HIP library (``ebm_cd_loss_f32``: fp64 block partials
    added in a fixed order; ``ebm_cd_loss_backward_f32``: the per-row seed) where autograd's graph of the same scalar arithmetic is
    some twenty launches of 4 - 5 us -- a tenth of a captured training step.  ``work``: the loss object's zeroed workspace."""

    @staticmethod
    def forward(ctx, e_both, n, reg, work):
        from .. import _lib

        e_both = e_both.contiguous()
        out = torch.empty(2, dtype=torch.float32, device=e_both.device)  # loss | finite flag
        _lib.call("ebm_cd_loss_f32", e_both.data_ptr(), n, float(reg), work.data_ptr(), out[0:1].data_ptr(), out[1:2].data_ptr(),
                  _lib.stream_handle(e_both.device))
        ctx.save_for_backward(e_both, out)
        ctx.n, ctx.reg = n, float(reg)
        return out[0]

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, g):
        from .. import _lib

        e_both, out = ctx.saved_tensors
        g = g.to(device=e_both.device, dtype=torch.float32).contiguous()
        seed = torch.empty_like(e_both)
        _lib.call("ebm_cd_loss_backward_f32", e_both.data_ptr(), ctx.n, ctx.reg, g.data_ptr(), out[1:2].data_ptr(), seed.data_ptr(),
                  _lib.stream_handle(e_both.device))
        return seed, None, None, None


class _PairedCDLoss(torch.autograd.Function):
    """The same loss and analytic gradient in torch ops (any dtype / device; five launches instead of autograd's twenty): what
    ``_PairedCDLossHip`` is checked against, and the route of energies that are not float32."""

    @staticmethod
    def forward(ctx, e_both, n, reg):
        e2 = e_both.view(2, n)
        means = e2.mean(dim=1)
        loss = means[0] - means[1]
        if reg > 0:
            loss = loss + reg * e2.square().mean(dim=1).sum()
        ok = torch.isfinite(loss)
        ctx.save_for_backward(e_both, ok)
        ctx.n, ctx.reg = n, reg
        return torch.where(ok, loss, loss.new_full((), 0.1))

    @staticmethod
    def backward(ctx, g):
        e_both, ok = ctx.saved_tensors
        n, reg = ctx.n, ctx.reg
        sign = e_both.new_tensor([1.0 / n, -1.0 / n]) if not e_both.is_cuda else torch.cat((e_both.new_full((1,), 1.0 / n), e_both.new_full((1,), -1.0 / n)))
        grad = sign.view(2, 1).expand(2, n)
        if reg > 0:
            grad = torch.addcmul(grad, e_both.view(2, n), e_both.new_full((), 2.0 * reg / n))
        return (grad * (g * ok)).reshape(-1), None, None


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
        # Two things this method knows and its callees cannot: `starts` is a fresh tensor nobody else holds (the sampler may run its
        # chains in it: no defensive copy of the state), and the weights do not change between the sampler call and the loss's model
        # call (an energy that packs its parameters for the kernels packs them once).  Each is a 4 us launch of a 700 us step.
        sampler, model = self.sampler, self.model
        # (storage, not identity: an overridden get_start_points may hand back a VIEW of the caller's batch)
        fresh = starts is not x and starts._base is None and (x.numel() == 0 or starts.untyped_storage().data_ptr() != x.untyped_storage().data_ptr())
        donate = getattr(sampler, "donate_input", None) is False and fresh
        scoped = hasattr(model, "_pack_scope") and model._pack_scope is None
        if donate:
            sampler.donate_input = True
        if scoped:
            model._pack_scope = {}
        try:
            negatives = sampler.sample(x=starts, n_steps=self.k_steps, model_kwargs=model_kwargs, generator=generator)
            if self.persistent:
                with torch.no_grad():
                    self.update_buffer(negatives)
            for key in self._CD_OPTION_KEYS:
                kwargs.setdefault(key, getattr(self, key))
            loss = self.compute_loss(x, negatives, *args, model_kwargs=model_kwargs, generator=generator, **kwargs)
        finally:
            if donate:
                sampler.donate_input = False
            if scoped:
                model._pack_scope = None
        return loss, negatives

    def _rows_independent(self) -> bool:
        """The model declares a row's energy independent of the batch around it AND is verifiably the network that declaration was
        made for: the class attribute is inherited by subclasses and survives a replaced ``net`` (train-mode BatchNorm, batch
        statistics), and one call on [data | negatives] is one forward-hook event where the reference makes two."""
        m = self.model
        if not getattr(m, "ROWS_INDEPENDENT", False):
            return False
        plain = getattr(m, "_plain_net", None)
        return bool(plain()) if callable(plain) else type(m).__dict__.get("ROWS_INDEPENDENT", False) is True

    def _paired_loss_work(self, device: torch.device) -> torch.Tensor:
        """The zeroed workspace of ``ebm_cd_loss_f32`` (every launch leaves it zeroed again); one per loss object and device."""
        from .. import _lib

        work = getattr(self, "_loss_work", None)
        if work is None or work.device != device:
            work = self._loss_work = torch.zeros(int(_lib.lib().ebm_cd_loss_work_bytes()) // 8 + 1, dtype=torch.float64, device=device)
        return work

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
            if (self._rows_independent() and not cond and real.is_cuda and real.shape == pred_x.shape
                    and not real.requires_grad and not pred_x.requires_grad):
                # one evaluation of both halves for an energy that declares every row's value independent of the batch around it
                # (MLPEnergy): the same arithmetic per row in half the launches -- the loss's forward / backward is some sixty
                # small kernels whose dispatch gaps, not their work, are a tenth of a captured training step
                e_both = self.model(torch.cat((real, pred_x)))
                # both halves' statistics from the [2, n] view: two row reductions instead of four means and two squares (every one
                # of these is a 10 us graph node on a 65 536-element vector)
                reg = float(kwargs.get("energy_reg_weight", self.energy_reg_weight))
                if e_both.dtype == torch.float32 and real.shape[0] > 0:
                    return _PairedCDLossHip.apply(e_both, int(real.shape[0]), reg, self._paired_loss_work(e_both.device))
                return _PairedCDLoss.apply(e_both, int(real.shape[0]), reg)
            e_data = self.model(real, **cond)
            e_model = self.model(pred_x, **cond)
        loss = torch.mean(e_data) - torch.mean(e_model)
        reg = kwargs.get("energy_reg_weight", self.energy_reg_weight)
        if reg > 0:
            loss = loss + reg * (torch.mean(e_data**2) + torch.mean(e_model**2))
        # a non-finite loss must not poison the optimiser: constant fallback, no host sync
        # (new_full: a fill kernel -- torch.tensor(0.1, device=...) is a pageable host-to-device copy, which a HIP graph cannot capture)
        return torch.where(torch.isfinite(loss), loss, loss.new_full((), 0.1))
