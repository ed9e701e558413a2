"""N > 1 path on CPU: world_size-2 gloo processes run the same control flow as bench.py
(rank -> clip range, constants published by rank 0, no data-path collective) with the CPU
oracle standing in for the device step; the concatenated shards must equal the unsharded
result bit for bit."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT, synth_clip


def _worker(rank, world, port, tmpdir):
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "zaf-python_amd"))
    import zafx
    from oracle import zaf_oracle as orc
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    # the 128-byte communicator id travels exactly like this in bench.py
    ids = [bytes(range(128)) if rank == 0 else None]
    dist.broadcast_object_list(ids, src=0)
    assert ids[0] == bytes(range(128))
    # constants are owned by rank 0 and broadcast (here through gloo; on GPUs through RCCL)
    import torch
    w = torch.from_numpy(zafx.hamming(256) if rank == 0 else np.zeros(256))
    dist.broadcast(w, src=0)
    n_clips, n = 7, 3000
    lo, hi = zafx.clip_range(n_clips, rank, world)
    shard = np.stack([synth_clip(4, c, n) for c in range(lo, hi)]).astype(np.float64)
    out = orc.stft_batch(shard, w.numpy(), 64)
    np.save(os.path.join(tmpdir, f"shard{rank}.npy"), out)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)   # the MAX-over-ranks timing reduction of bench.py
    assert t.item() == world
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_clip_sharding(tmp_path):
    import torch.multiprocessing as mp
    from oracle import zaf_oracle as orc
    import zafx
    world, port = 2, 29500 + os.getpid() % 2000
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    got = np.concatenate([np.load(tmp_path / f"shard{r}.npy") for r in range(world)])
    full = np.stack([synth_clip(4, c, 3000) for c in range(7)]).astype(np.float64)
    ref = orc.stft_batch(full, zafx.hamming(256), 64)
    assert got.shape == ref.shape and np.array_equal(got, ref)
