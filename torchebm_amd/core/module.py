"""``TorchEBMModule``: the ``nn.Module`` base every component derives from.

Host-side mirror of the reference's torchebm/core/base_module.py (device/dtype probe
:63-97, conditioning normalisation :105-141, autocast shim :143-176, ``warn_once`` :33).
Pure plumbing -- nothing here is on the kernel path.
"""

from __future__ import annotations

import contextlib
import warnings
from typing import Any, Dict, Optional, Union

import torch
from torch import nn

_already_warned: set = set()
_WARNED_ONCE = _already_warned  # the reference's name for the registry (core/base_module.py:30; its tests clear it)


def warn_once(key: str, message: str, category: type = DeprecationWarning, stacklevel: int = 3) -> None:
    """``warnings.warn`` at most once per process for ``key`` (hot loops must not re-warn)."""
    if key not in _already_warned:
        _already_warned.add(key)
        warnings.warn(message, category, stacklevel=stacklevel)


def _canonical(device: torch.device) -> torch.device:
    # "cuda:0" and "cuda" compare unequal in torch; the library treats them as one.
    if device.type == "cuda" and device.index == 0:
        return torch.device("cuda")
    return device


_normalize = _canonical  # the reference's name for it (core/base_module.py:24-27)


class TorchEBMModule(nn.Module):
    """``nn.Module`` whose ``device`` / ``dtype`` follow its parameters (or, for a
    parameter-less module, a zero-element probe buffer that ``.to()`` moves along)."""

    def __init__(
        self,
        device: Union[str, torch.device, None] = None,
        dtype: Optional[torch.dtype] = None,
        *args: Any,
        **kwargs: Any,
    ):
        super().__init__(*args, **kwargs)
        probe = torch.empty(0, dtype=dtype or torch.get_default_dtype(), device=device)
        self.register_buffer("_torchebm_probe", probe, persistent=False)
        self.use_mixed_precision = False
        self.autocast_available = False
        self._amp_dtype = torch.float16
        self._where: Optional[tuple] = None  # (device, dtype) cache, dropped by _apply

    def _locate(self) -> tuple:
        if self._where is None:
            anchor = next(self.parameters(), None)
            if anchor is None:
                anchor = self._torchebm_probe
            self._where = (_canonical(anchor.device), anchor.dtype)
        return self._where

    @property
    def device(self) -> torch.device:
        return self._locate()[0]

    @property
    def dtype(self) -> torch.dtype:
        return self._locate()[1]

    def _apply(self, fn, recurse: bool = True):
        out = super()._apply(fn, recurse=recurse)
        self._where = None
        return out

    def _prepare_model_kwargs(self, model_kwargs: Optional[dict]) -> Dict[str, Any]:
        """Normalise conditioning once per call: tensors go to ``self.device`` (no dtype
        cast -- labels stay integral), everything else passes through; always a new dict."""
        if not model_kwargs:
            return {}
        if not isinstance(model_kwargs, dict):
            raise TypeError(f"model_kwargs must be a dict, got {type(model_kwargs).__name__}")
        dev = self.device
        out = {}
        for key, val in model_kwargs.items():
            out[key] = val.to(dev, non_blocking=True) if torch.is_tensor(val) else val
        return out

    def setup_mixed_precision(self, use_mixed_precision: bool, amp_dtype: torch.dtype = torch.float16) -> None:
        self.use_mixed_precision = bool(use_mixed_precision)
        self._amp_dtype = amp_dtype
        self.autocast_available = False
        if not self.use_mixed_precision:
            return
        if self.device.type != "cuda":
            warnings.warn(
                f"Mixed precision requested but device is {self.device}. Requires CUDA. "
                "Falling back to full precision.",
                UserWarning,
            )
            self.use_mixed_precision = False
            return
        self.autocast_available = True

    def autocast_context(self):
        if self.use_mixed_precision and self.autocast_available:
            return torch.amp.autocast(device_type=self.device.type, dtype=self._amp_dtype)
        return contextlib.nullcontext()


def graph_state_key(model) -> tuple:
    """What a captured HIP graph of ``model``'s forward / backward has frozen, as a hashable key: the storage of
    every parameter and buffer (in-place updates -- an optimiser step -- are seen by a replay, a REPLACED tensor is
    not) and every plain Python attribute of every submodule (``training``, a temperature, a flag: a replay cannot
    see a changed value).  The samplers re-capture when the key of the model they are about to replay differs."""
    import torch

    items = []
    for name, mod in model.named_modules():
        for attr, val in vars(mod).items():
            if attr.startswith("_"):
                continue
            if isinstance(val, (bool, int, float, str, type(None))):
                items.append((name, attr, val))
            elif isinstance(val, (tuple, list)) and all(isinstance(v, (bool, int, float, str)) for v in val):
                items.append((name, attr, tuple(val)))
        items.append((name, "training", mod.training))
    tensors = tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in list(model.parameters()) + list(model.buffers()))
    return (tuple(items), tensors)
