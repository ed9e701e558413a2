"""Clip-level sharding across the GPUs of one node.

Clips are independent, so the batch is split into contiguous blocks, one per rank
(one process per GPU).  There is no data-path collective: the only communication is
the RCCL broadcast of the shared constants at plan creation (core.Comm).
"""

__all__ = ["clip_range", "shard_sizes", "run_sharded"]


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


def run_sharded(batch_fn, clips, devices, *args, **kwargs):
    """Run a batched transform (`stft_batch`, `mdct_batch`, `melspectrogram_batch`, ...) with the
    clip axis block-partitioned over `devices`, one host thread per device (ctypes releases the GIL
    during libzafx calls; every device has its own plan and stream).  Single process: each plan gets
    its constants from the host, so no collective is needed; results are concatenated by index.

        X = zafx.run_sharded(zafx.stft_batch, clips, range(zafx.device_count()), window, 1024)
    """
    import threading

    import numpy as np

    devices = list(devices)
    if not devices:
        raise ValueError("devices must not be empty")
    clips = np.asarray(clips)
    if clips.ndim < 2:
        raise ValueError("clips must have a leading clip axis")
    ranges = [clip_range(clips.shape[0], r, len(devices)) for r in range(len(devices))]
    results = [None] * len(devices)
    errors = []

    def work(r):
        lo, hi = ranges[r]
        try:
            if hi > lo:
                results[r] = batch_fn(clips[lo:hi], *args, device=devices[r], **kwargs)
        except Exception as exc:   # re-raised in the caller: the whole batch fails if any device fails
            errors.append(exc)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(len(devices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        raise errors[0]
    parts = [p for p in results if p is not None]
    if not parts:
        return batch_fn(clips, *args, device=devices[0], **kwargs)
    return np.concatenate(parts, axis=0)
