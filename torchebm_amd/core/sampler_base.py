"""``BaseSampler``: the API contract shared by the MCMC samplers.

Host-side mirror of the reference's torchebm/core/base_sampler.py (:11-155).  The
``sample`` signature (parameter order, defaults, keyword-only ``model_kwargs`` /
``generator``) and the return contract are pinned by the reference's
tests/samplers/test_api_contract.py:27-46 and reproduced by every sampler here.

Diagnostics contract (``return_diagnostics=True`` -> ``(tensor, dict)``), values of shape
``[n_kept, ...]`` with ``n_kept = n_steps // thin``:  ``"mean"``, ``"var"`` (biased, clamped
to [1e-10, 1e10]), ``"energy"`` and, for HMC, ``"acceptance_rate"``.
"""

from __future__ import annotations

from abc import ABC, abstractmethod
from typing import Dict, Optional, Tuple, Union

import torch
from torch import nn

from .module import TorchEBMModule
from .schedules import Schedulable


class BaseSampler(Schedulable, TorchEBMModule, ABC):
    def __init__(
        self,
        model: nn.Module,
        dtype: torch.dtype = torch.float32,
        device: Optional[Union[str, torch.device]] = None,
    ):
        super().__init__(device=device, dtype=dtype)
        self.model = model

    def _init_state(
        self,
        x: Optional[torch.Tensor],
        dim: Optional[Union[int, Tuple[int, ...]]],
        n_samples: int,
        generator: Optional[torch.Generator] = None,
    ) -> torch.Tensor:
        """The caller's ``x`` on the sampler's device/dtype, or ``n_samples`` draws from N(0, I)."""
        if x is not None:
            return x.to(device=self.device, dtype=self.dtype)
        if dim is None:
            raise ValueError("dim must be provided when x is None")
        shape = (dim,) if isinstance(dim, int) else tuple(dim)
        return torch.randn(n_samples, *shape, dtype=self.dtype, device=self.device, generator=generator)

    # conditioning convention: an empty dict means "call the model exactly as before"
    def _model_gradient(self, x: torch.Tensor, model_kwargs: Dict[str, object]) -> torch.Tensor:
        if model_kwargs:
            return self.model.gradient(x, model_kwargs=model_kwargs)
        return self.model.gradient(x)

    def _model_energy(self, x: torch.Tensor, model_kwargs: Dict[str, object]) -> torch.Tensor:
        if model_kwargs:
            return self.model(x, **model_kwargs)
        fast = getattr(self.model, "_sampler_energy", None)  # (a kernel pass for the sampler's own energy calls: GaussianModel)
        if fast is not None and not torch.is_grad_enabled():
            e = fast(x)
            if e is not None:
                return e
        return self.model(x)

    @abstractmethod
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
        generator: Optional[torch.Generator] = None,
    ) -> Union[torch.Tensor, Tuple[torch.Tensor, Dict[str, torch.Tensor]]]:
        """Run the sampler; see the concrete classes for the produced diagnostics."""
        raise NotImplementedError
