"""bench.py's N-rank plumbing on the GPU box every round (housekeeping: no multi-GPU box has been in reach, so what CAN run runs):
  * ZAFX_BENCH_FORCE_DIST=1 -- one rank through the whole distributed flow: file rendezvous, RCCL communicator (ncclCommInitRank through the
    dlopen'd library), broadcast of the plan constants, barriers, MAX over ranks; the line must carry rccl.ranks_seen = 1 and no error;
  * `--gpus 8` self-launched on the one GPU (ZAFX_BENCH_SHARE_DEVICES=1): eight rank processes, rendezvous, id hand-off, per-rank shards and
    timing reduction -- RCCL refuses eight ranks on one device, so the communicator is allowed to fail (ZAFX_BENCH_ALLOW_NO_COMM=1) and the
    line must say so in its `error` field while every rank still ran its share.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def run_bench(extra_env, *args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ZAFX_RDZV_DIR", "ZAFX_RDZV_NS")}
    env.update(extra_env)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--kind", "stft", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", *args],
                         env=env, capture_output=True, text=True, timeout=timeout)
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (res.returncode, res.stdout[-1000:], res.stderr[-2000:])
    return res.returncode, json.loads(lines[0])


def test_one_rank_through_the_distributed_flow():
    rc, line = run_bench({"ZAFX_BENCH_FORCE_DIST": "1", "ZAFX_BENCH_PREWARM_S": "0.1"})
    assert rc == 0 and "error" not in line, line
    assert line["n_gpus"] == 1 and line["rccl"]["ok"] and line["rccl"]["ranks_seen"] == [1]   # (what ncclCommCount reported on every rank)
    assert line["parity"]["within_tolerance"] and line["value"] > 0


def test_eight_ranks_self_launched_on_one_gpu():
    rc, line = run_bench({"ZAFX_BENCH_SHARE_DEVICES": "1", "ZAFX_BENCH_ALLOW_NO_COMM": "1", "ZAFX_BENCH_COMM_TIMEOUT": "60", "ZAFX_BENCH_PREWARM_S": "0.1"},
                         "--gpus", "8", timeout=900)
    assert rc == 0, line
    assert line["n_gpus"] == 8 and len(line["rccl"]["kernel_ms_per_rank"]) == 8 and all(t > 0 for t in line["rccl"]["kernel_ms_per_rank"])
    assert line["config"]["parallelism"] == "clip-sharded x8" and line["parity"]["within_tolerance"]
    if not line["rccl"]["ok"]:   # (eight ranks on one device: RCCL says no, and the line says that it did)
        assert "error" in line and "RCCL" in line["error"]
