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


_BLIND = object()
_MODULE_INTERNALS = None


def _plain(val):
    """A hashable stand-in for an attribute value a captured graph may have frozen, or ``_BLIND``."""
    import torch

    if isinstance(val, (bool, int, float, complex, str, bytes, type(None), torch.device, torch.dtype, torch.Size)):
        return val
    if torch.is_tensor(val):
        return ("tensor", val.data_ptr(), val._version, tuple(val.shape), val.dtype)
    if isinstance(val, (tuple, list)):
        parts = tuple(_plain(v) for v in val)
        return _BLIND if any(p is _BLIND for p in parts) else (type(val).__name__,) + parts
    if isinstance(val, dict):
        try:
            parts = tuple(sorted((str(k), _plain(v)) for k, v in val.items()))
        except TypeError:
            return _BLIND
        return _BLIND if any(p[1] is _BLIND for p in parts) else ("dict",) + parts
    if isinstance(val, type) or callable(val) and not isinstance(val, torch.nn.Module) and not hasattr(val, "__dict__"):
        return ("callable", id(val))
    import types

    if isinstance(val, (types.FunctionType, types.BuiltinFunctionType, types.MethodType)):
        return ("callable", id(val))
    return _BLIND


def _module_attributes(model):
    """(module name, attribute name, value) of every attribute a user put on a submodule (torch's own bookkeeping --
    ``_parameters``, hooks ... -- is skipped: parameters and buffers enter the key through their storages)."""
    import torch

    global _MODULE_INTERNALS
    if _MODULE_INTERNALS is None:
        _MODULE_INTERNALS = frozenset(vars(torch.nn.Module()).keys())
    for name, mod in model.named_modules():
        for attr, val in vars(mod).items():
            if attr in _MODULE_INTERNALS or attr == "_where":  # _where: DeviceMixin's own (device, dtype) cache
                continue
            yield name, attr, val


def graph_state_key(model) -> tuple:
    """What a captured HIP graph of ``model``'s forward / backward has frozen, as a hashable key: the storage of
    every parameter and buffer (in-place updates -- an optimiser step -- are seen by a replay, a REPLACED tensor is
    not); every Python attribute of every submodule, ``_``-prefixed ones included -- numbers, strings, flags, tuples /
    lists / dicts of those, functions by identity, and tensors kept as plain attributes (``self.scale = torch.tensor(..)``:
    storage, in-place version, shape); ``training``; and whether autocast is on.  The samplers re-capture when the key of
    the model they are about to replay differs.  Attributes it cannot hash are listed by ``graph_blind_spots``."""
    import torch

    items = []
    for name, attr, val in _module_attributes(model):
        k = _plain(val)
        items.append((name, attr, "<unhashable>" if k is _BLIND else k))  # the PRESENCE of a blind spot is part of the key
    for name, mod in model.named_modules():
        items.append((name, "training", mod.training))
    tensors = tuple((t.data_ptr(), tuple(t.shape), t.dtype) for t in list(model.parameters()) + list(model.buffers()))
    return (tuple(items), tensors, torch.is_autocast_enabled())


def graph_blind_spots(model) -> list:
    """Attributes ``graph_state_key`` cannot see into (an arbitrary object, a container holding one): a replay would not
    notice a change behind them, so the samplers do not capture BY DEFAULT when a model has any (``capture_graph = True``
    overrides)."""
    return [f"{name or type(model).__name__}.{attr}" for name, attr, val in _module_attributes(model) if _plain(val) is _BLIND]


class ForwardProbe:
    """What ONE eager iteration of the step route did besides computing: taken around the first (real) step of a call,
    it tells whether the model's forward has side effects a replayed graph would not reproduce the way eager launches
    do -- a buffer written under ``forward`` (BatchNorm running statistics in training mode), random numbers drawn from
    the default CUDA or CPU generator (dropout, ``torch.rand`` in the forward), a Python attribute that changed (a call
    counter).  The samplers capture by default only when it reports nothing."""

    def __init__(self, model, device):
        import torch

        self.model, self.device = model, device
        self.buffers = tuple(b._version for b in model.buffers())
        self.cuda_offset = self._cuda_offset()
        self.cpu_state = torch.get_rng_state()
        self.key = graph_state_key(model)

    def _cuda_offset(self):
        import torch

        if self.device.type != "cuda":
            return None
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        return torch.cuda.default_generators[idx].get_offset()

    def side_effects(self) -> list:
        import torch

        found = []
        if tuple(b._version for b in self.model.buffers()) != self.buffers:
            found.append("the forward writes a buffer (running statistics in training mode?)")
        if self._cuda_offset() != self.cuda_offset:
            found.append("the forward draws from the default CUDA generator (dropout?)")
        if not torch.equal(torch.get_rng_state(), self.cpu_state):
            found.append("the forward draws from the default CPU generator")
        if graph_state_key(self.model) != self.key:
            found.append("the forward changes an attribute of the model")
        return found


