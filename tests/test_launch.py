"""N > 1 path on CPU, without torch.distributed: bench.py's own launcher and file rendezvous (zafx/launch.py).

`bench.py --gpus 2 --selftest-launch` runs the rank control flow of the real bench (rendezvous from the environment,
broadcast of the 128-byte communicator id, clip-range sharding, barrier, MAX over ranks) with the CPU oracle standing
in for the device step; the shards' checksums must add up to the unsharded result.  Both ways in are driven: the
self-launcher (`python bench.py --gpus 2`, WORLD_SIZE unset) and the driver's `python -m torch.distributed.run ...`.
"""
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

from conftest import ROOT, synth_clip


def _expected():
    from oracle import zaf_oracle as orc
    import zafx
    full = np.stack([synth_clip(4, c, 3000) for c in range(7)]).astype(np.float64)
    ref = orc.stft_batch(full, orc.hamming_periodic(256), 64)
    return [(lo, hi, float(np.sum(np.abs(ref[lo:hi])))) for lo, hi in (zafx.clip_range(7, r, 2) for r in range(2))]


def _check(line):
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["uid_ok"] and rec["max_rank_plus_1"] == 2.0
    for got, want in zip(rec["shards"], _expected()):
        assert (int(got[0]), int(got[1])) == want[:2]
        assert got[2] == want[2]   # the same NumPy calls on the same clips: bit-identical checksums


def _clean_env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "ZAFX_RDZV_DIR")}
    return env


@pytest.mark.timeout(120)
def test_self_launcher_two_ranks():
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"], env=_clean_env(),
                         capture_output=True, timeout=100)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    _check(res.stdout.decode().strip().splitlines()[-1])


@pytest.mark.timeout(180)
def test_torchrun_two_ranks():
    port = 29600 + os.getpid() % 2000
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--selftest-launch"]
    res = subprocess.run(cmd, env=_clean_env(), capture_output=True, timeout=170)
    assert res.returncode == 0, res.stderr.decode()[-2000:]
    lines = [ln for ln in res.stdout.decode().splitlines() if ln.startswith("{")]
    assert len(lines) == 1   # rank 0 prints ONE JSON line
    _check(lines[0])


def test_rendezvous_collectives(tmp_path):
    from zafx.launch import Rendezvous
    world, results, errors = 3, {}, []

    def rank_main(r):
        try:
            rv = Rendezvous(str(tmp_path / "rv"), r, world, timeout=30.0)
            uid = rv.broadcast(b"\x01" * 128 if r == 0 else b"")
            rv.barrier()
            mx = rv.all_reduce_max([float(r), 10.0 - r])
            parts = rv.all_gather(bytes([r]))
            rv.close()
            results[r] = (uid, mx, parts)
        except Exception as exc:   # pragma: no cover
            errors.append(exc)

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors
    for r in range(world):
        uid, mx, parts = results[r]
        assert uid == b"\x01" * 128 and mx == [2.0, 10.0] and parts == [b"\x00", b"\x01", b"\x02"]
    assert not os.path.exists(tmp_path / "rv")   # rank 0 removed the directory after the last rank said goodbye


def test_rendezvous_rejects_bad_rank(tmp_path):
    from zafx.launch import Rendezvous
    with pytest.raises(ValueError):
        Rendezvous(str(tmp_path), 2, 2)


def test_bench_does_not_import_torch():
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "import torch" not in src
    for name in ("launch.py", "core.py", "shard.py", "_lib.py", "__init__.py", "constants.py"):
        assert "import torch" not in open(os.path.join(ROOT, "zaf-python_amd", "zafx", name)).read()


def test_rendezvous_directory_names_the_launcher_instance(monkeypatch):
    """Ranks started by one launcher derive the same directory; it carries the launcher's pid AND start time, so a later
    launcher that is handed the same pid (fresh boxes start counting low) cannot meet the files of a job that died."""
    from zafx import launch
    monkeypatch.delenv("ZAFX_RDZV_DIR", raising=False)
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("MASTER_PORT", "29999")
    a, b = launch.Rendezvous.from_env(), launch.Rendezvous.from_env()
    try:
        assert a.dir == b.dir
        assert a.dir.endswith(f"_{os.getppid()}_{launch._start_time(os.getppid())}")
        assert launch._start_time(os.getppid()) > 0 and launch._start_time(2 ** 30) == 0
    finally:
        a.close()


def test_rendezvous_fails_fast_on_a_multi_node_launch(monkeypatch):
    """global RANK / WORLD_SIZE with a node-local directory blocked for the whole timeout."""
    from zafx import launch
    monkeypatch.setenv("RANK", "3")
    monkeypatch.setenv("WORLD_SIZE", "16")
    monkeypatch.setenv("LOCAL_WORLD_SIZE", "8")
    with pytest.raises(RuntimeError, match="ONE node"):
        launch.Rendezvous.from_env()


def test_rendezvous_directory_is_private_and_stale_files_are_ignored(tmp_path, monkeypatch):
    from zafx import launch
    d = tmp_path / "shared"
    d.mkdir(mode=0o777)
    os.chmod(d, 0o777)
    with pytest.raises(PermissionError):          # a directory other users can write to is refused
        launch.Rendezvous(str(d), 0, 1)
    os.chmod(d, 0o755)                            # (made under the usual umask -- nobody else can write: accepted)
    (d / "000001_bcast").write_bytes(b"stale id of a job that died")   # what a reused ZAFX_RDZV_DIR may hold
    monkeypatch.setenv("ZAFX_RDZV_DIR", str(d))
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.delenv("LOCAL_WORLD_SIZE", raising=False)
    rv = launch.Rendezvous.from_env(timeout=5.0)
    assert rv.ns and rv.broadcast(b"fresh") == b"fresh"
    assert rv.get("000001_bcast") == b"fresh"     # this job's key, not the stale file of the same number
    rv.close()
    assert (d / "000001_bcast").exists()          # another job's file is not ours to delete
    # ranks with different parent processes (started by hand) meet through an explicit namespace
    monkeypatch.setenv("ZAFX_RDZV_NS", "job42")
    rv = launch.Rendezvous.from_env(timeout=5.0)
    # (the key prefix is a fixed-length digest of the name -- ADVICE r5: the name itself made "job42." a prefix of job "job42.5"'s keys, and
    # close() deletes by prefix)
    import hashlib
    tag = lambda name: hashlib.blake2b(name.encode(), digest_size=8).hexdigest() + "."
    assert rv.base_ns == tag("job42") and rv.ns.startswith(rv.base_ns) and len(rv.ns) > len(rv.base_ns) + 8   # (+ this launch's epoch)
    monkeypatch.setenv("ZAFX_RDZV_NS", "job42.5")
    other = launch.Rendezvous.from_env(timeout=5.0)
    other.put("alive", b"a running job's key")
    assert not other.base_ns.startswith(rv.base_ns) and not rv.base_ns.startswith(other.base_ns)
    rv.close()                                    # job42 leaves ...
    assert other.get("alive") == b"a running job's key"   # ... and job42.5's keys are still there
    rv = other
    monkeypatch.delenv("ZAFX_RDZV_NS")
    monkeypatch.setenv("TORCHELASTIC_RUN_ID", "run7")
    assert launch.Rendezvous.from_env(timeout=5.0).base_ns == tag("run7")
    rv.put("secret", b"unique id")
    assert os.stat(d / (rv.ns + "secret")).st_mode & 0o077 == 0   # key files are private whatever the directory allows


@pytest.mark.timeout(60)
def test_named_namespace_survives_a_job_that_died(tmp_path):
    """a relaunch under the same ZAFX_RDZV_NS / TORCHELASTIC_RUN_ID in a reused directory must not read the RCCL id
    (or the handshake files) a dead job left there."""
    from zafx import launch
    d = tmp_path / "reused"
    d.mkdir(mode=0o700)
    for name, data in (("job.000001_bcast", b"stale id"), ("job.hello_1", b"00ff"), ("job.epoch", b"dead 00ff"), ("job.ack_1", b"dead"),
                       ("job.dead.000001_bcast", b"stale id of the epoch before")):
        (d / name).write_bytes(data)
    got = {}

    def rank(r):
        rv = launch.Rendezvous(str(d), r, 2, timeout=30.0, namespace="job.")
        rv.handshake()
        got[r] = (rv.ns, rv.broadcast(b"fresh id" if r == 0 else b""))
        rv.close()

    ts = [threading.Thread(target=rank, args=(r,)) for r in (1, 0)]
    ts[0].start()
    import time
    time.sleep(0.2)     # rank 1 is already waiting, with the stale epoch file in front of it, when rank 0 arrives
    ts[1].start()
    for t in ts:
        t.join(40)
    assert got[0][0] == got[1][0] and "dead" not in got[0][0]
    assert got[0][1] == got[1][1] == b"fresh id"
    assert not d.exists() or not [n for n in os.listdir(d) if n.startswith("job.")]   # close() took the dead job's files of this name along


@pytest.mark.timeout(60)
def test_spawn_ranks_stops_the_others_when_a_rank_dies(tmp_path):
    """rank 1 crashing left rank 0 waiting out the rendezvous timeout."""
    import time
    from zafx import launch
    script = tmp_path / "ranks.py"
    script.write_text("import os, sys, time\nif os.environ['RANK'] == '1':\n    sys.exit(7)\ntime.sleep(300)\n")
    t0 = time.monotonic()
    code, out = launch.spawn_ranks([str(script)], 2)
    assert code == 7 and time.monotonic() - t0 < 30
