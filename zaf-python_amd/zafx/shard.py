"""Clip-level sharding across the GPUs of one node.

Clips are independent, so the batch is split into contiguous blocks, one per rank
(one process per GPU).  There is no data-path collective: the only communication is
the RCCL broadcast of the shared constants at plan creation (core.Comm).
"""

__all__ = ["clip_range", "shard_sizes"]


def clip_range(n_clips, rank, world_size):
    """Half-open clip interval owned by `rank`: [floor(r*B/G), floor((r+1)*B/G))."""
    if world_size < 1 or not 0 <= rank < world_size:
        raise ValueError("bad rank / world_size")
    if n_clips < 0:
        raise ValueError("n_clips must be >= 0")
    return (rank * n_clips) // world_size, ((rank + 1) * n_clips) // world_size


def shard_sizes(n_clips, world_size):
    """Number of clips per rank (sums to n_clips, differs by at most one)."""
    return [b - a for a, b in (clip_range(n_clips, r, world_size) for r in range(world_size))]
