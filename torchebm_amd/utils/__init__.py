from .distributed import (
    all_gather_cat,
    all_reduce_diagnostics,
    broadcast_object,
    broadcast_tensor,
    get_rank,
    get_world_size,
    is_distributed,
    sample_and_gather,
    shard_rows,
    unsharded,
)

from .graphed_step import GraphedTrainingStep

__all__ = ["GraphedTrainingStep", "all_gather_cat", "all_reduce_diagnostics", "broadcast_object", "broadcast_tensor", "get_rank", "get_world_size", "is_distributed", "sample_and_gather", "shard_rows", "unsharded"]
