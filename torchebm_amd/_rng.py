"""Host-side view of a ``torch.Generator``'s Philox state for the HIP kernels.

The kernels draw from a Philox4x32-10 stream addressed by ``(seed, step, element)``
(see ``include/ebm_hip.h``).  ``seed`` is the generator's seed and ``step`` starts at
``philox_offset // 4``; after a call that consumed ``n`` steps the generator's offset is
advanced by ``4 * n``.  All of this is host bookkeeping on the generator object -- no
device read, no synchronisation (the reference's Triton POC drew its seed with
``torch.randint(...).item()``, a host sync: cuda/fused_langevin.py:121-122).

This preserves the generator contract of the reference's tests/test_generator.py:74-113:
same seed => identical samples, different seeds differ, ``generator=None`` consumes the
device's default generator, an explicit generator leaves the default one untouched.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch


def _resolve(generator: Optional[torch.Generator], device: torch.device) -> torch.Generator:
    if generator is None:
        idx = device.index if device.index is not None else torch.cuda.current_device()
        return torch.cuda.default_generators[idx]
    if generator.device.type != device.type:
        # same failure class as torch.randn(..., device=cuda, generator=<cpu generator>)
        raise RuntimeError(
            f"Expected a '{device.type}' device type for generator but found '{generator.device.type}'"
        )
    return generator


def _get_offset(gen: torch.Generator) -> int:
    if hasattr(gen, "get_offset"):
        return int(gen.get_offset())
    state = gen.get_state()  # CUDA generator state: 8 bytes seed + 8 bytes offset (host tensor)
    return int.from_bytes(bytes(state[8:16].tolist()), "little")


def _set_offset(gen: torch.Generator, offset: int) -> None:
    if hasattr(gen, "set_offset"):
        gen.set_offset(offset)
        return
    state = gen.get_state().clone()
    state[8:16] = torch.tensor(list(int(offset).to_bytes(8, "little")), dtype=torch.uint8)
    gen.set_state(state)


#: XORed into the high word of the Philox key.  torch's own CUDA draws from the same generator use
#: key = seed with counter (offset/4, 0, thread, 0); the kernels' counter space (element group, step)
#: overlaps that index space, so with the bare seed as key e.g. `randn`'s thread `tid` at offset 0 and the
#: kernel's group 0 at step `tid` would read the same 128 random bits.  A distinct key is an independent
#: Philox stream.  ("EBM1")
STREAM_TAG = 0x45424D31


def kernel_seed(seed: int) -> int:
    """The Philox key the kernels use for a generator seeded with ``seed`` (tests use it to materialise
    the field a sampler call drew with ``ebm_noise_fill_f32``)."""
    return (int(seed) ^ (STREAM_TAG << 32)) & 0xFFFFFFFFFFFFFFFF


def reserve(generator: Optional[torch.Generator], device: torch.device, n_steps: int) -> Tuple[int, int]:
    """Return ``(seed, first_step)`` for a kernel call that consumes ``n_steps`` RNG steps
    and advance the generator past them."""
    gen = _resolve(generator, device)
    seed = kernel_seed(gen.initial_seed())
    offset = _get_offset(gen)
    _set_offset(gen, offset + 4 * int(n_steps))
    return seed, offset // 4


class DeviceCoords:
    """RNG coordinates in DEVICE memory for launches captured in a HIP graph (``utils.graphed_step``): ``tensor`` is
    ``int64[2] = {kernel seed, step}`` as raw 64-bit patterns; a launch inside the capture asks ``take(n)`` for its
    offset from ``step`` (the ``step_delta`` argument of the ``*_dev_f32`` entry points) -- the deltas are handed out in
    program order, exactly as ``reserve`` would have advanced the generator between the same calls."""

    def __init__(self, device: torch.device):
        self.tensor = torch.zeros(2, dtype=torch.int64, device=device)
        self.taken = 0

    def take(self, n_steps: int) -> int:
        delta = self.taken
        self.taken += int(n_steps)
        return delta

    @staticmethod
    def as_i64(v: int) -> int:
        """The int64 with the bit pattern of the uint64 ``v``."""
        return v - (1 << 64) if v >= (1 << 63) else v
