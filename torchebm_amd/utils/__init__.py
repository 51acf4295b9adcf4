from .distributed import all_gather_cat, broadcast_tensor, get_rank, get_world_size, is_distributed, shard_rows

__all__ = ["all_gather_cat", "broadcast_tensor", "get_rank", "get_world_size", "is_distributed", "shard_rows"]
