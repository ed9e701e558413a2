"""One process per GPU without a framework: rank discovery, a file rendezvous and a self-launcher.

The hot path shards by clips and has no data-path collective, so the only things the ranks of
one node must exchange are small host values: the 128-byte RCCL unique id (rank 0 -> everyone, so that
`Comm` can broadcast the plan constants over xGMI), a barrier either side of the timed region and the
MAX over ranks of the elapsed time.  That is what this module provides, with files in a directory every
rank of the node can see -- no torch.distributed, no MPI, no sockets to collide with the launcher's own.

Two ways in:
  * launched by `python -m torch.distributed.run ...` (or any launcher that sets RANK / LOCAL_RANK /
    WORLD_SIZE / MASTER_ADDR / MASTER_PORT): `Rendezvous.from_env()` derives the directory from
    MASTER_PORT and the launcher's pid (all ranks of one node share the parent process);
  * plain `python script.py --gpus N` with WORLD_SIZE unset: `spawn_ranks()` starts the N ranks itself
    (environment as above plus ZAFX_RDZV_DIR) and relays rank 0's stdout.
"""
import hashlib
import os
import struct
import subprocess
import sys
import tempfile
import time

__all__ = ["Rendezvous", "rank_env", "spawn_ranks"]


def rank_env():
    """(rank, local_rank, world_size) from the launcher's environment; (0, 0, 1) when there is none."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", os.environ.get("RANK", "0"))),
            int(os.environ.get("WORLD_SIZE", "1")))


def _single_node_or_raise():
    """The rendezvous directory is local to one node: a launch over several nodes (WORLD_SIZE above LOCAL_WORLD_SIZE) would
    wait for files that can never appear, so it fails here, at once."""
    world, local = int(os.environ.get("WORLD_SIZE", "1")), os.environ.get("LOCAL_WORLD_SIZE")
    if local is not None and int(local) != world:
        raise RuntimeError(f"zafx.launch: the file rendezvous serves the ranks of ONE node (WORLD_SIZE = {world}, "
                           f"LOCAL_WORLD_SIZE = {local}); run one job per node -- the clips shard without any exchange")


def _start_time(pid):
    """Start time of a process in clock ticks since boot (field 22 of /proc/<pid>/stat); 0 where /proc is not there."""
    try:
        with open(f"/proc/{pid}/stat", "rb") as f:
            return int(f.read().rsplit(b")", 1)[1].split()[19])
    except (OSError, ValueError, IndexError):
        return 0


class Rendezvous:
    """Key/value exchange, barrier and MAX-reduce between the ranks of one node through a shared directory.

    Every operation is named by the caller or numbered in call order; all ranks must make the same
    sequence of collective calls (as with any collective library).  Values are published by an atomic
    rename, so a reader never sees a partial file."""

    def __init__(self, directory, rank, world_size, timeout=600.0, namespace=""):
        if world_size < 1 or not 0 <= rank < world_size:
            raise ValueError("bad rank / world_size")
        self.dir, self.rank, self.world, self.timeout = directory, int(rank), int(world_size), float(timeout)
        self.ns = str(namespace)   # prefix of every key: the files of an earlier job in a reused directory are never read
        self.base_ns = self.ns     # (handshake() appends this launch's epoch to ns)
        self._seq = 0
        # private to this user: another local user must not be able to create the directory first and plant keys in it
        # (the RCCL unique id travels through here)
        os.makedirs(self.dir, mode=0o700, exist_ok=True)
        st = os.stat(self.dir)
        # (group / other WRITE access is what lets someone else plant keys; a directory made under the usual umask, 0755, is fine)
        if st.st_uid != os.getuid() or (st.st_mode & 0o022):
            raise PermissionError(f"rendezvous directory {self.dir} is not writable by uid {os.getuid()} alone "
                                  f"(owner {st.st_uid}, mode {st.st_mode & 0o777:o})")

    @classmethod
    def from_env(cls, timeout=600.0):
        _single_node_or_raise()
        rank, _, world = rank_env()
        d = os.environ.get("ZAFX_RDZV_DIR")
        if not d:
            # all ranks of a node are children of one launcher process: its pid + the rendezvous port name the job
            # (+ the launcher's start time: a pid can come round again on a fresh box, and a job that died leaves its files behind)
            key = f"{os.environ.get('MASTER_PORT', '0')}_{os.environ.get('TORCHELASTIC_RUN_ID', 'none')}_{os.getppid()}_{_start_time(os.getppid())}"
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
            d = os.path.join(base, f"zafx_rdzv_{os.getuid()}_{key}")
        # a directory the caller names may be reused: the keys carry the launcher's identity (all ranks of a node are children
        # of one launcher process), so files a previous job left there are not this job's
        # -- unless the job names itself (ZAFX_RDZV_NS, or the launcher's TORCHELASTIC_RUN_ID): ranks started by hand or from
        # per-rank wrapper shells have different parents and would otherwise never see each other's keys
        ns = os.environ.get("ZAFX_RDZV_NS") or os.environ.get("TORCHELASTIC_RUN_ID")
        named = bool(ns) and ns != "none"
        if not named:
            ns = f"{os.getppid()}.{_start_time(os.getppid())}"
        # the prefix of every key is a fixed-length digest of the name, not the name: "job42." is a prefix of "job42.5.<epoch>.bcast", and
        # close() deletes by prefix -- one job's close would take a running job's keys with it
        rv = cls(d, rank, world, timeout, namespace=hashlib.blake2b(ns.encode(), digest_size=8).hexdigest() + ".")
        if named:
            # a NAME comes back with every relaunch (an elastic restart keeps its run id): the keys of a job that died before
            # close() are still there under it, so the ranks first agree on an epoch that only this launch knows
            rv.handshake()
        return rv

    def _read(self, name):
        try:
            with open(os.path.join(self.dir, name), "rb") as f:
                return f.read()
        except FileNotFoundError:
            return None

    def handshake(self):
        """Agree on a per-launch epoch under a namespace that earlier launches may have used: rank r publishes a
        fresh nonce, rank 0 publishes a fresh epoch together with the nonces it has read, rank r accepts only an epoch file
        that carries ITS nonce and acknowledges with the epoch, rank 0 waits until every acknowledgement carries ITS epoch.
        Whatever an earlier launch left under the same names holds other random values and is never accepted; afterwards
        every key is prefixed with the epoch."""
        base, others = self.base_ns, range(1, self.world)
        deadline = time.monotonic() + self.timeout

        def wait(delay):
            if time.monotonic() > deadline:
                raise TimeoutError(f"rendezvous: rank {self.rank} waited {self.timeout:.0f} s for the handshake under '{base}' in {self.dir}")
            time.sleep(delay)
            return min(delay * 2, 0.02)

        delay = 0.0005
        if self.rank == 0:
            epoch, seen = os.urandom(8).hex(), None
            while self.world > 1:
                nonces = [self._read(f"{base}hello_{r}") for r in others]
                if all(n is not None for n in nonces):
                    if nonces != seen:
                        self._write(f"{base}epoch", b" ".join([epoch.encode()] + nonces))
                        seen = nonces
                    if all(self._read(f"{base}ack_{r}") == epoch.encode() for r in others):
                        break
                delay = wait(delay)
        else:
            nonce = os.urandom(8).hex().encode()
            self._write(f"{base}hello_{self.rank}", nonce)
            while True:
                parts = (self._read(f"{base}epoch") or b"").split(b" ")
                if len(parts) == self.world and parts[self.rank] == nonce:
                    epoch = parts[0].decode()
                    self._write(f"{base}ack_{self.rank}", parts[0])
                    break
                delay = wait(delay)
        self.ns = f"{base}{epoch}."

    # ---- point to point -------------------------------------------------------------
    def _write(self, name, data):
        """Publish one file by an atomic rename; created private (0600) whatever the umask and the directory's mode are: the
        RCCL unique id (bootstrap address + magic) travels through here and is no other local user's to read."""
        tmp = os.path.join(self.dir, f".{name}.{self.rank}.tmp")
        try:
            os.unlink(tmp)
        except FileNotFoundError:
            pass
        fd = os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600)
        with os.fdopen(fd, "wb") as f:
            f.write(bytes(data))
        os.replace(tmp, os.path.join(self.dir, name))

    def put(self, key, data):
        self._write(self.ns + key, data)

    def get(self, key):
        path = os.path.join(self.dir, self.ns + key)
        deadline = time.monotonic() + self.timeout
        delay = 0.0005
        while True:
            try:
                with open(path, "rb") as f:
                    return f.read()
            except FileNotFoundError:
                if time.monotonic() > deadline:
                    raise TimeoutError(f"rendezvous: rank {self.rank} waited {self.timeout:.0f} s for '{key}' in {self.dir}") from None
                time.sleep(delay)
                delay = min(delay * 2, 0.02)

    # ---- collectives ----------------------------------------------------------------
    def _next(self, what):
        self._seq += 1
        return f"{self._seq:06d}_{what}"

    def broadcast(self, data, root=0):
        """`data` (bytes) of rank `root` returned on every rank."""
        key = self._next("bcast")
        if self.rank == root:
            self.put(key, data)
            return bytes(data)
        return self.get(key)

    def all_gather(self, data):
        key = self._next("gather")
        self.put(f"{key}_{self.rank}", data)
        return [self.get(f"{key}_{r}") for r in range(self.world)]

    def barrier(self):
        self.all_gather(b"")

    def all_reduce_max(self, values):
        """Element-wise MAX of a sequence of floats over the ranks (the timing reduction of bench.py)."""
        values = [float(v) for v in values]
        parts = self.all_gather(struct.pack(f"<{len(values)}d", *values))
        rows = [struct.unpack(f"<{len(values)}d", p) for p in parts]
        return [max(col) for col in zip(*rows)] if values else []

    def close(self):
        """Last call of every rank: rank 0 removes the directory once every other rank has said it will not read again."""
        self.barrier()
        if self.rank != 0:
            self.put(f"bye_{self.rank}", b"")
            return
        for r in range(1, self.world):
            self.get(f"bye_{r}")
        for name in os.listdir(self.dir):
            # (base_ns: with a named namespace also what earlier launches of this name left behind -- one job per name at a time)
            if not (name.startswith(self.base_ns) or name.startswith("." + self.base_ns)):
                continue   # (another job's files in a shared directory)
            try:
                os.unlink(os.path.join(self.dir, name))
            except OSError:
                pass
        try:
            os.rmdir(self.dir)   # (fails, harmlessly, while another job's files are there)
        except OSError:
            pass


def spawn_ranks(argv, n_ranks, env=None, timeout=None):
    """Start `n_ranks` copies of `python argv...`, rank r bound to GPU r (RANK = LOCAL_RANK = r), wait for them and
    return (exit code, rank 0's stdout).  The children find each other through ZAFX_RDZV_DIR."""
    if n_ranks < 1:
        raise ValueError("n_ranks must be >= 1")
    rdzv = tempfile.mkdtemp(prefix="zafx_rdzv_", dir="/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None)
    procs = []
    try:
        for r in range(n_ranks):
            e = dict(os.environ if env is None else env)
            e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n_ranks), LOCAL_WORLD_SIZE=str(n_ranks), ZAFX_RDZV_DIR=rdzv,
                     MASTER_ADDR=e.get("MASTER_ADDR", "127.0.0.1"), MASTER_PORT=e.get("MASTER_PORT", "29400"))
            e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # RCCL between processes needs dmabuf IPC on this driver
            procs.append(subprocess.Popen([sys.executable] + list(argv), env=e, stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
        # rank 0's stdout is drained on a thread while ALL children are polled: a rank that dies takes the others down with
        # it at once (they would wait out the rendezvous timeout for its files)
        import threading
        chunks = []
        reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
        reader.start()
        deadline = None if timeout is None else time.monotonic() + timeout
        code = 0
        while True:
            states = [p.poll() for p in procs]
            bad = [c for c in states if c not in (None, 0)]
            if bad:
                code = bad[0]
                break
            if all(c == 0 for c in states):
                break
            if deadline is not None and time.monotonic() > deadline:
                raise subprocess.TimeoutExpired([sys.executable] + list(argv), timeout)
            time.sleep(0.02)
        if code:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        for p in procs:
            p.wait()
        reader.join(timeout=10)
        return code, b"".join(c for c in chunks if c).decode()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
        for name in os.listdir(rdzv) if os.path.isdir(rdzv) else []:
            try:
                os.unlink(os.path.join(rdzv, name))
            except OSError:
                pass
        try:
            os.rmdir(rdzv)
        except OSError:
            pass
