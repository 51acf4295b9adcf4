"""Guarded collectives (reference: torchebm/utils/distributed.py:21-125).

The sampler itself never communicates: chains are independent, each rank owns a contiguous
block of rows and seeds its generator ``base_seed + rank``.  The one exchange on the path is
the read-back of the sharded final state -- ``all_gather_cat`` -- which on ROCm's "nccl"
backend is an RCCL all-gather over xGMI (one equal-sized shard per rank, rank-ordered).
Every helper is the identity when ``torch.distributed`` is not initialised.
"""

from __future__ import annotations

import contextlib
from typing import Iterator, Optional

import torch
import torch.distributed as dist


def is_distributed() -> bool:
    return dist.is_available() and dist.is_initialized()


def get_world_size(group=None) -> int:
    return dist.get_world_size(group) if is_distributed() else 1


def get_rank(group=None) -> int:
    return dist.get_rank(group) if is_distributed() else 0


def all_gather_cat(x: torch.Tensor, group=None) -> torch.Tensor:
    """Concatenate ``x`` from every rank along dim 0, in rank order.  Shapes must match
    across ranks.  Uses a single ``all_gather_into_tensor`` (one large message per peer
    instead of ``world`` list entries) -- the shape that suits point-to-point xGMI links."""
    world = get_world_size(group)
    if world == 1:
        return x
    x = x.contiguous()
    out = torch.empty((world * x.shape[0],) + tuple(x.shape[1:]), dtype=x.dtype, device=x.device)
    dist.all_gather_into_tensor(out, x, group=group)
    return out


def sample_and_gather(sampler, x: torch.Tensor, n_steps: int, *, pieces: Optional[int] = None, generator=None, group=None):
    """This rank's shard through ``sampler.sample`` AND the read-back all-gather, pipelined: the shard
    is processed in ``pieces`` row blocks, and the all-gather of block i is in flight (RCCL's own
    stream, ``async_op=True``) while block i+1 is being sampled, so only the last block's gather is
    exposed.  Each block is still tens of thousands of workgroups at the sizes this matters for.

    Returns ``(local, gathered)``: ``local`` is this rank's final ``[n, ...]`` state; ``gathered`` is a
    view of shape ``[world, pieces, n // pieces, ...]`` -- ``gathered[r]`` are rank r's chains in order
    (block-major storage; ``.reshape(world * n, ...)`` materialises the rank-ordered concatenation).
    The rank-ordered concatenation is one device copy away (4 GiB at BASELINE config-4 size: ~1.6 ms at the
    chip's copy rate, against 282 ms of sampling per call).
    With one process the same ``pieces`` ``sampler.sample`` calls are made (so a rank's chains do not depend
    on how many other ranks exist) and ``gathered`` is the ``[1, pieces, n // pieces, ...]`` view of the
    local result -- the indexing contract does not depend on the world size.  An explicit ``pieces`` must divide
    ``n``; the default (``None``) takes the largest of 4, 3, 2, 1 that does, so any shard size works.
    Every block runs at the SAME point of the sampler's schedules (step size, noise scale ...): the schedulers are
    rewound before each block and end up advanced by ``n_steps`` once, as after one ``sample()`` call over the shard
    -- ``pieces`` is a pipelining choice, not a change of the dynamics."""
    n = x.shape[0]
    world = get_world_size(group)
    if pieces is None:
        pieces = next(p for p in (4, 3, 2, 1) if n % p == 0)
    if pieces < 1 or n % pieces != 0:
        raise ValueError(f"n = {n} chains cannot be split into {pieces} equal blocks")
    block = n // pieces
    tail = tuple(x.shape[1:])
    local = torch.empty_like(x)
    scheds = list(sampler._subtree_schedulers()) if hasattr(sampler, "_subtree_schedulers") else []
    start = [s.state_dict() for s in scheds]

    def sample_block(i):
        if i > 0:
            for s, st in zip(scheds, start):
                s.load_state_dict(st)
        return sampler.sample(x=x[i * block : (i + 1) * block], n_steps=n_steps, generator=generator)

    if world == 1:
        for i in range(pieces):
            local[i * block : (i + 1) * block] = sample_block(i)
        return local, local.view((1, pieces, block) + tail)
    store = torch.empty((pieces, world, block) + tail, dtype=x.dtype, device=x.device)
    pending = []
    for i in range(pieces):
        out = sample_block(i)
        local[i * block : (i + 1) * block] = out
        pending.append(dist.all_gather_into_tensor(store[i].view((world * block,) + tail), out.contiguous(),
                                                   group=group, async_op=True))
    for work in pending:
        work.wait()
    return local, store.transpose(0, 1)


def all_reduce_diagnostics(diag: dict, n_local: int, group=None) -> dict:
    """Combine per-rank sampler diagnostics (``mean`` / ``var`` ``[n_kept, dim]``, ``energy`` and, for HMC,
    ``acceptance_rate`` ``[n_kept]``, each over this rank's ``n_local`` chains) into the statistics of the
    whole sharded population: ONE all-reduce of the count-weighted first and second moments (a few
    ``dim``-sized vectors -- negligible next to the sampling).  Identity with one process."""
    world = get_world_size(group)
    if world == 1:
        return diag
    w = float(n_local)
    keys = [k for k in ("energy", "acceptance_rate") if k in diag]
    parts = [diag["mean"] * w, (diag["var"] + diag["mean"] ** 2) * w] + [diag[k].unsqueeze(-1) * w for k in keys]
    flat = torch.cat([p.reshape(p.shape[0], -1) for p in parts] + [torch.full_like(diag["energy"], w).unsqueeze(-1)], dim=1)
    flat = flat.to(torch.float64)
    dist.all_reduce(flat, group=group)
    total = flat[:, -1:]
    d = diag["mean"].reshape(diag["mean"].shape[0], -1).shape[1]
    mean = flat[:, :d] / total
    var = (flat[:, d : 2 * d] / total - mean**2).clamp_(min=1e-10, max=1e10)
    out = {"mean": mean.to(diag["mean"].dtype).view_as(diag["mean"]), "var": var.to(diag["var"].dtype).view_as(diag["var"])}
    for i, k in enumerate(keys):
        out[k] = (flat[:, 2 * d + i] / total[:, 0]).to(diag[k].dtype)
    return out


def broadcast_object(obj, src: int = 0, group=None):
    """A picklable object (a seed, a config dict) from rank ``src`` to every rank; the object itself when not
    distributed (reference: utils/distributed.py:73-96).  Control-plane only: the data path of a sharded sampling
    run has no collective before the read-back."""
    if get_world_size(group) == 1:
        return obj
    box = [obj if get_rank(group) == src else None]
    dist.broadcast_object_list(box, src=src, group=group)
    return box[0]


def broadcast_tensor(x: torch.Tensor, src: int = 0, group=None) -> torch.Tensor:
    """Broadcast ``x`` from ``src``; a CPU tensor hops through the GPU when the backend is
    NCCL/RCCL (which cannot move host memory)."""
    if get_world_size(group) == 1:
        return x
    backend = dist.get_backend(group)
    if backend == "nccl" and not x.is_cuda:
        dev = torch.device("cuda", torch.cuda.current_device())
        y = x.to(dev)
        dist.broadcast(y, src=src, group=group)
        return y.to(x.device)
    dist.broadcast(x, src=src, group=group)
    return x


def shard_rows(n_total: int, group=None) -> tuple:
    """``(start, count)`` of this rank's contiguous block of ``n_total`` chains; the first
    ``n_total % world`` ranks take one extra row."""
    world, rank = get_world_size(group), get_rank(group)
    base, extra = divmod(n_total, world)
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


@contextlib.contextmanager
def unsharded(module: torch.nn.Module, recurse: bool = True) -> Iterator[torch.nn.Module]:
    """Hold an FSDP2-wrapped energy model's parameters gathered for the duration of a sampling block
    (reference: utils/distributed.py:129-175): a k-step chain calls the model k times, and with reshard-after-forward
    each call would repeat the parameter all-gather.  Duck-typed on ``set_reshard_after_forward`` -- plain modules, DDP
    and single-process runs pass through untouched; nothing is imported from FSDP and no process group is needed."""
    toggle = getattr(module, "set_reshard_after_forward", None)
    if toggle is None:
        yield module
        return
    toggle(False, recurse=recurse)
    try:
        yield module
    finally:
        toggle(True, recurse=recurse)
