"""Loss bases: ``BaseLoss`` and ``BaseContrastiveDivergence`` (replay buffer).

Host-side mirror of the reference's torchebm/core/base_loss.py:22-114 (BaseLoss) and
:116-530 (BaseContrastiveDivergence).  This is the *caller* of the sampler hot path
(SURVEY.md §8 row C): it only needs ``sampler.sample(x=, n_steps=, model_kwargs=,
generator=)`` to hand back a detached ``[B, ...]`` tensor.  All of its own arithmetic is
plain PyTorch.
"""

from __future__ import annotations

import logging
import warnings
from abc import ABC, abstractmethod
from typing import Any, Optional, Tuple, Union

import torch

from .. import _lib, _rng
from .module import TorchEBMModule, warn_once
from .schedules import Schedulable

logger = logging.getLogger(__name__)


class BaseLoss(Schedulable, TorchEBMModule, ABC):
    def __init__(
        self,
        dtype: torch.dtype = torch.float32,
        device: Optional[Union[str, torch.device]] = None,
        *args: Any,
        **kwargs: Any,
    ):
        super().__init__(*args, device=device, dtype=dtype, **kwargs)

    def _resolve_model_kwargs(self, model_kwargs: Optional[dict], legacy_kwargs: Optional[dict] = None, *, warn_key: str) -> dict:
        if legacy_kwargs:
            warn_once(
                warn_key,
                "Passing model conditioning as bare keyword arguments is deprecated; pass model_kwargs={...} instead.",
            )
            model_kwargs = {**legacy_kwargs, **(model_kwargs or {})}
        return self._prepare_model_kwargs(model_kwargs)

    @abstractmethod
    def forward(self, x: torch.Tensor, *args, **kwargs) -> torch.Tensor: ...

    def __repr__(self) -> str:
        return f"{type(self).__name__}()"

    __str__ = __repr__


class BaseContrastiveDivergence(BaseLoss):
    """CD machinery shared by the CD variants: where negative chains start (data, or a
    persistent replay buffer with stratified reads and FIFO writes)."""

    def __init__(
        self,
        model,
        sampler,
        k_steps: int = 1,
        persistent: bool = False,
        buffer_size: int = 100,
        new_sample_ratio: float = 0.0,
        init_steps: int = 0,
        dtype: torch.dtype = torch.float32,
        device: Optional[Union[str, torch.device]] = None,
        *args,
        **kwargs,
    ):
        super().__init__(*args, dtype=dtype, device=device, **kwargs)
        self.model = model
        self.sampler = sampler
        self.k_steps = k_steps
        self.persistent = persistent
        self.buffer_size = buffer_size
        self.new_sample_ratio = new_sample_ratio
        self.init_steps = init_steps
        # both live in state_dict once the lazy initialisation has happened
        self.register_buffer("replay_buffer", None)
        self.register_buffer("buffer_ptr", torch.tensor(0, dtype=torch.long, device=self.device))
        self._write_pos = 0  # host copy of buffer_ptr: the FIFO never reads the device scalar
        self.buffer_initialized = False
        #: set by ``utils.graphed_step.GraphedTrainingStep`` while it captures: the buffer kernels then take their RNG
        #: coordinates and the write position from device memory (``_rng.DeviceCoords``, ``buffer_ptr``)
        self._graph_coords = None
        self._graph_fifo_rows = 0  # rows the captured step appends to the FIFO per replay (0: whole-buffer overwrite or no buffer)

    # ---- replay buffer ------------------------------------------------------------------
    def initialize_buffer(
        self,
        data_shape_no_batch: Tuple[int, ...],
        buffer_chunk_size: int = 1024,
        init_noise_scale: float = 0.01,
        generator: Optional[torch.Generator] = None,
    ) -> Optional[torch.Tensor]:
        """Fill the buffer with small Gaussian noise, then (optionally) burn it in with
        ``init_steps`` sampler steps, chunk by chunk (base_loss.py:190-264)."""
        if not self.persistent or self.buffer_initialized:
            return None
        if self.buffer_size <= 0:
            raise ValueError(f"Replay buffer size must be positive, got {self.buffer_size}")
        shape = (self.buffer_size,) + tuple(data_shape_no_batch)
        logger.info("Initializing replay buffer with shape %s...", shape)
        self.replay_buffer = torch.randn(shape, dtype=self.dtype, device=self.device, generator=generator) * init_noise_scale
        if self.init_steps > 0:
            chunk = min(self.buffer_size, buffer_chunk_size)
            with torch.no_grad():
                for lo in range(0, self.buffer_size, chunk):
                    hi = min(lo + chunk, self.buffer_size)
                    seed_rows = self.replay_buffer[lo:hi].clone()
                    try:
                        with self.autocast_context():
                            burned = self.sampler.sample(x=seed_rows, n_steps=self.init_steps, generator=generator).detach()
                    except Exception as exc:  # keep the noise for this chunk, like the reference
                        warnings.warn(f"Error during buffer initialization sampling for chunk {lo}-{hi}: {exc}. Keeping noise for this chunk.")
                        continue
                    if burned.shape == seed_rows.shape:
                        self.replay_buffer[lo:hi] = burned
                    else:
                        warnings.warn(
                            f"Sampler output shape mismatch during buffer init. Expected {seed_rows.shape}, "
                            f"got {burned.shape}. Skipping update for chunk {lo}-{hi}."
                        )
        self.buffer_ptr.zero_()
        self._write_pos = 0
        self.buffer_initialized = True
        return self.replay_buffer

    #: Opt-in (an attribute; round 6): draw the exploration subset of a persistent-CD step as the REFERENCE does
    #: -- ``randperm(batch)[:n_new]`` and ``randn`` on the caller's torch generator (core/base_loss.py:316-332) -- instead of the one-launch
    #: form (``ebm_pcd_start_points_f32``: exactly ``n_new`` rows through a keyed Feistel bijection, uniform over subsets to the tests'
    #: resolution, normals from the kernels' Philox field).  The same marginal law either way; ``True`` costs a device sort and eight small
    #: launches per step (0.13 ms of a 0.7 ms config-5 step) and cannot be part of a captured training step (``GraphedTrainingStep``
    #: refuses it).  The stratified gather itself stays one HIP launch.
    reference_subset: bool = False

    def get_start_points(self, x: torch.Tensor, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        """Chain starts: a copy of the data (CD) or stratified reads of the buffer (PCD:
        one row per stride with a random in-stride offset, base_loss.py:266-337)."""
        x = x.to(device=self.device, dtype=self.dtype)
        batch = x.shape[0]
        if not self.persistent:
            return x.detach().clone()
        if not self.buffer_initialized:
            self.initialize_buffer(tuple(x.shape[1:]), generator=generator)
            if not self.buffer_initialized:
                raise RuntimeError("Buffer initialization failed.")
        if self.buffer_size < batch:
            warnings.warn(
                f"Buffer size ({self.buffer_size}) is smaller than batch size ({batch}). Sampling with replacement.",
                UserWarning,
            )
            rows = torch.randint(0, self.buffer_size, (batch,), device=self.device, generator=generator)
            starts = self.replay_buffer[rows]
        elif self._hip_buffer():
            # one launch: in-kernel Philox offsets + gather (ebm_pcd_gather_f32)
            stride = self.buffer_size // batch
            row_elems = self.replay_buffer[0].numel()
            starts = torch.empty((batch,) + tuple(self.replay_buffer.shape[1:]), dtype=self.dtype, device=self.device)
            if self.new_sample_ratio > 0.0 and batch < 2**31 and not self.reference_subset:
                # ... and the exploration noise in the same launch (ebm_pcd_start_points_f32): the random subset of n_new rows comes
                # from a keyed bijection instead of randperm's sort, the normals from the kernels' Philox field -- no torch-side draw,
                # so a graph-captured step replays the eager loop's numbers (utils.graphed_step)
                n_new = max(1, int(batch * self.new_sample_ratio))
                if self._graph_coords is not None:
                    seed, step, coords = 0, self._graph_coords.take(3), _lib.ptr(self._graph_coords.tensor)
                else:
                    (seed, step), coords = _rng.reserve(generator, self.replay_buffer.device, 3), None
                _lib.call(
                    "ebm_pcd_start_points_f32", _lib.ptr(self.replay_buffer), self.buffer_size, row_elems, _lib.ptr(starts), batch, stride,
                    n_new, 0.01, seed, step, coords, _lib.stream_handle(self.replay_buffer.device),
                )
                return starts
            if self._graph_coords is not None:  # being captured: the draw's step is read from device memory at every replay
                _lib.call(
                    "ebm_pcd_gather_dev_f32", _lib.ptr(self.replay_buffer), self.buffer_size, row_elems, _lib.ptr(starts), batch,
                    stride, None, _lib.ptr(self._graph_coords.tensor), self._graph_coords.take(1),
                    _lib.stream_handle(self.replay_buffer.device),
                )
            else:
                seed, step = _rng.reserve(generator, self.replay_buffer.device, 1)
                _lib.call(
                    "ebm_pcd_gather_f32", _lib.ptr(self.replay_buffer), self.buffer_size, row_elems, _lib.ptr(starts), batch,
                    stride, None, None, seed, step, _lib.stream_handle(self.replay_buffer.device),
                )
        else:
            stride = self.buffer_size // batch
            base = torch.arange(0, batch, device=self.device) * stride
            jitter = torch.randint(0, stride, (batch,), device=self.device, generator=generator)
            rows = (base + jitter) % self.buffer_size
            starts = self.replay_buffer[rows]
        if self.new_sample_ratio > 0.0:
            n_new = max(1, int(batch * self.new_sample_ratio))
            pick = torch.randperm(batch, device=self.device, generator=generator)[:n_new]
            bump = torch.randn_like(starts[pick], device=self.device, dtype=self.dtype, generator=generator) * 0.01
            starts[pick] = starts[pick] + bump
        return starts

    def _hip_buffer(self) -> bool:
        """The replay buffer can be served by the HIP gather / scatter kernels."""
        buf = self.replay_buffer
        return buf is not None and buf.is_cuda and buf.dtype == torch.float32 and buf.is_contiguous()

    def get_negative_samples(self, x, batch_size, data_shape, generator: Optional[torch.Generator] = None) -> torch.Tensor:
        kw = dict(dtype=self.dtype, device=self.device)
        if not self.persistent or not self.buffer_initialized:
            return torch.randn((batch_size,) + tuple(data_shape), generator=generator, **kw)
        n_new = max(1, int(batch_size * self.new_sample_ratio))
        n_old = batch_size - n_new
        out = torch.empty((batch_size,) + tuple(data_shape), **kw)
        out[:n_new] = torch.randn((n_new,) + tuple(data_shape), generator=generator, **kw)
        if n_old > 0:
            rows = torch.randint(0, self.buffer_size, (n_old,), device=self.device, generator=generator)
            out[n_new:] = self.replay_buffer[rows]
        return out

    @property
    def _buffer_ptr_int(self) -> int:
        """The reference's name for the host copy of ``buffer_ptr`` (core/base_loss.py:187; its tests read it)."""
        return self._write_pos

    def update_buffer(self, samples: torch.Tensor) -> None:
        """FIFO write with wrap-around; the write position is tracked on the host
        (base_loss.py:390-426)."""
        if not self.persistent or not self.buffer_initialized:
            return
        samples = samples.to(device=self.device, dtype=self.dtype).detach()
        batch, cap, pos = samples.shape[0], self.buffer_size, self._write_pos
        if batch >= cap:
            self.replay_buffer[:] = samples[-cap:]
            new_pos = 0
        elif self._hip_buffer() and samples.is_cuda and self._graph_coords is not None:
            # being captured: the write position is the DEVICE scalar, advanced inside the graph; the host copy follows
            # arithmetically after every replay (GraphedTrainingStep -> _graph_replayed)
            src = _lib.dense_f32(samples)
            _lib.call(
                "ebm_pcd_scatter_dev_f32", _lib.ptr(self.replay_buffer), cap, self.replay_buffer[0].numel(), _lib.ptr(src),
                batch, _lib.ptr(self.buffer_ptr), _lib.stream_handle(self.replay_buffer.device),
            )
            self.buffer_ptr.add_(batch).remainder_(cap)
            self._graph_fifo_rows = batch
            return
        elif self._hip_buffer() and samples.is_cuda:
            new_pos = (pos + batch) % cap
            src = _lib.dense_f32(samples)
            _lib.call(
                "ebm_pcd_scatter_f32", _lib.ptr(self.replay_buffer), cap, self.replay_buffer[0].numel(), _lib.ptr(src),
                batch, pos, _lib.stream_handle(self.replay_buffer.device),
            )
        else:
            new_pos = (pos + batch) % cap
            if new_pos > pos:
                self.replay_buffer[pos:new_pos] = samples
            else:
                head = cap - pos
                self.replay_buffer[pos:] = samples[:head]
                self.replay_buffer[:new_pos] = samples[head:]
        self._write_pos = new_pos
        self.buffer_ptr.fill_(new_pos)

    def _graph_replayed(self) -> None:
        """Host bookkeeping after one replay of a captured training step: the FIFO position the graph advanced on the device."""
        if self._graph_fifo_rows:
            self._write_pos = (self._write_pos + self._graph_fifo_rows) % self.buffer_size

    def mix_buffer_across_ranks(self, process_group=None, generator: Optional[torch.Generator] = None) -> None:
        """Shuffle the union of all ranks' buffers with one shared permutation and keep this
        rank's slice: an exact partition of the union (base_loss.py:428-481)."""
        if not self.persistent:
            raise RuntimeError("mix_buffer_across_ranks requires a persistent loss (persistent=True).")
        if not self.buffer_initialized:
            raise RuntimeError(
                "The replay buffer is not initialized; run one training step or call initialize_buffer() first."
            )
        from ..utils.distributed import all_gather_cat, broadcast_tensor, get_rank, get_world_size

        if get_world_size(process_group) == 1:
            return
        union = all_gather_cat(self.replay_buffer, group=process_group)
        perm = broadcast_tensor(torch.randperm(union.shape[0], generator=generator), src=0, group=process_group)
        lo = get_rank(process_group) * self.buffer_size
        self.replay_buffer.copy_(union[perm[lo : lo + self.buffer_size].to(union.device)])

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)
        self._write_pos = int(self.buffer_ptr.item())

    @abstractmethod
    def forward(self, x: torch.Tensor, *args, **kwargs) -> Tuple[torch.Tensor, torch.Tensor]: ...

    @abstractmethod
    def compute_loss(self, x: torch.Tensor, pred_x: torch.Tensor, *args, **kwargs) -> torch.Tensor: ...

    def __repr__(self) -> str:
        return f"{type(self).__name__}(model={self.model}, sampler={self.sampler})"

    __str__ = __repr__
