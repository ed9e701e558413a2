"""Host-side logic of the product (no GPU): constant builders against the golden vectors
from the reference, argument validation, sharding arithmetic, the FFT core emulated on CPU."""
import os
import subprocess

import numpy as np
import pytest
import scipy.fftpack
import scipy.signal.windows as sw
import scipy.sparse

import zafx
from conftest import ROOT, relerr


def csr(g, tag):
    return scipy.sparse.csr_matrix((g[f"{tag}_data"], g[f"{tag}_indices"], g[f"{tag}_indptr"]), shape=tuple(g[f"{tag}_shape"]))


def test_melfilterbank_matches_reference(golden):
    for tag, args in (("fb128", (44100, 2048, 128)), ("fb40", (44100, 2048, 40))):
        ref = csr(golden["consts"], tag)
        got = zafx.melfilterbank(*args)
        assert scipy.sparse.issparse(got) and got.shape == ref.shape and got.nnz == ref.nnz
        assert np.array_equal(got.toarray(), ref.toarray())
    assert np.array_equal(zafx.melfilterbank(8000, 64, 8).toarray(), golden["tiny"]["fb_dense"])


def test_cqtkernel_matches_reference(golden):
    ref = csr(golden["consts"], "ck_small")
    got = zafx.cqtkernel(4000, 12, 200, 1600)
    assert got.shape == ref.shape and got.nnz == ref.nnz
    assert relerr(got.toarray(), ref.toarray()) <= 1e-12


@pytest.mark.timeout(120)
def test_cqtkernel_config_matches_reference(golden):
    ref = csr(golden["consts"], "ck")
    got = zafx.cqtkernel(44100, 24, 55, 3520).tocsr()
    got.sort_indices()
    assert got.shape == (144, 32768) and got.nnz == 9450
    assert np.array_equal(got.indices, ref.indices) and relerr(got.data, ref.data) <= 1e-12


@pytest.mark.timeout(180)
def test_cqtkernel_docstring_example_matches_reference(golden):
    """zaf.py:476-483 (55 Hz ... fs/2) through the product's host builder; and the plan routing for it."""
    g = golden["cqtfull"]
    got = zafx.cqtkernel(44100, 24, 55, 44100 / 2).tocsr()
    got.sort_indices()
    assert got.shape == (208, 32768) and np.array_equal(got.indptr, g["indptr"])
    assert np.array_equal(got.indices[g["probe_idx"]], g["probe_col"])
    assert relerr(got.data[g["probe_idx"]], g["probe_val"]) <= 1e-12


def test_dct_rows_match_scipy():
    y = np.random.default_rng(0).standard_normal((128, 7))
    ref = scipy.fftpack.dct(y, axis=0, norm="ortho")[1:21]
    assert relerr(zafx.dct2_rows(128, 20) @ y, ref) <= 1e-12


def test_windows():
    assert np.allclose(zafx.hamming(2048), sw.hamming(2048, sym=False), atol=1e-15)
    assert np.allclose(zafx.hamming(64, periodic=False), sw.hamming(64, sym=True), atol=1e-15)
    assert np.allclose(zafx.kaiser_bessel_derived(2048), sw.kaiser_bessel_derived(2048, beta=5 * np.pi), atol=1e-14)
    s = zafx.sine(64)
    assert np.max(np.abs(s[:32] ** 2 + s[32:] ** 2 - 1)) < 1e-14


def test_argument_validation_happens_before_the_device():
    x = np.zeros(5000, np.float32)
    ham = zafx.hamming(2048)
    with pytest.raises(ValueError):
        zafx.stft(np.zeros((2, 5000)), ham, 1024)
    with pytest.raises(ValueError):
        zafx.stft(x, ham, 1024.0)
    with pytest.raises(ValueError):
        zafx.stft(x, ham, 0)
    with pytest.raises(ValueError):
        zafx.stft(x, zafx.hamming(9000), 500)
    with pytest.raises(ValueError):
        zafx.istft(np.zeros((2048, 4), complex), ham, 4096)   # (the forward transforms take a hop above the window, as zaf.stft)
    with pytest.raises(ValueError):
        zafx.stft(x.astype(complex), ham, 1024)
    with pytest.raises(ValueError):
        zafx.istft(np.zeros(2048, complex), ham, 1024)
    with pytest.raises(ValueError):
        zafx.istft(np.zeros((1024, 4), complex), ham, 1024)
    with pytest.raises(ValueError):
        zafx.melspectrogram(x, ham, 1024, np.ones((4, 1024)))
    with pytest.raises(ValueError):
        zafx.mfcc(x, ham, 1024, zafx.melfilterbank(44100, 2048, 40), 40)
    with pytest.raises(ValueError):
        zafx.imdct(np.zeros((512, 4)), ham)
    with pytest.raises(ValueError):
        zafx.cqtspectrogram(x, 44100, 25, np.ones((4, 512)))
    # spectrum kinds and precision: checked on the host too
    with pytest.raises(ValueError):
        zafx.stft_batch(x[None], ham, 1024, onesided="phase")
    with pytest.raises(ValueError):
        zafx.istft_batch(np.zeros((1, 1025, 4), complex), ham, 1024, onesided="power")
    with pytest.raises(ValueError):
        zafx.istft_batch(np.zeros((1, 2048, 4), complex), ham, 1024, onesided=True)   # rows must be W/2 + 1
    with pytest.raises(ValueError):
        zafx.set_precision("f16")
    assert zafx.get_precision() == "f32"


def test_shard_partition():
    for b in (0, 1, 7, 8, 1024, 8191, 8192):
        for g in (1, 2, 3, 4, 8):
            sizes = zafx.shard_sizes(b, g)
            assert sum(sizes) == b and max(sizes) - min(sizes) <= 1
            edges = [zafx.clip_range(b, r, g) for r in range(g)]
            assert edges[0][0] == 0 and edges[-1][1] == b
            assert all(edges[r][1] == edges[r + 1][0] for r in range(g - 1))
    assert zafx.clip_range(8192, 3, 8) == (3072, 4096)   # BASELINE config 5: 1024 clips per GPU
    with pytest.raises(ValueError):
        zafx.clip_range(10, 2, 2)


@pytest.mark.timeout(300)
def test_fft_core_emulated_on_cpu(tmp_path):
    """zafx_fft.hpp (pass schedule, padded LDS indexing, twiddle tables, register DFTs) compiled
    for the host with threads emulated by loops, against a float64 DFT."""
    exe = tmp_path / "fft_emu"
    subprocess.run(["g++", "-O2", "-std=c++17", "-DZAFX_HOST_EMU", "-I", os.path.join(ROOT, "zaf-python_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host_emu", "fft_emu.cpp"), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout
    worst = float(out.strip().splitlines()[-1].split("=")[1])
    assert worst < 5e-7, out


@pytest.mark.timeout(300)
def test_mel64_tables_emulated_on_cpu(tmp_path):
    """zafx_mel64.hpp (the float64 mel kernel's view of the filterbank: its non-zeros as one equally long stream per lane + partial sums)
    compiled for the host, the kernel's product loop run lane by lane, against the dense product -- for the reference's filterbanks and
    for random bands, an all-zero row, two filters and a dense matrix."""
    import struct
    exe = tmp_path / "mel64_emu"
    subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "zaf-python_amd", "csrc"),
                    os.path.join(ROOT, "tests", "host_emu", "mel64_emu.cpp"), "-o", str(exe)], check=True)
    rng = np.random.default_rng(5)
    banks = [zafx.melfilterbank(fs, 2048, nf).toarray() for fs, nf in ((44100, 128), (44100, 40), (16000, 64), (44100, 2), (8000, 13))]
    rnd = np.zeros((128, 1024))
    for r in range(128):
        a = int(rng.integers(0, 1000))
        b = min(1024, a + int(rng.integers(1, 200)))
        rnd[r, a:b] = rng.standard_normal(b - a)
    rnd[5] = 0.0
    banks += [rnd, rng.standard_normal((16, 1024))]
    for fb in banks:
        s = rng.standard_normal(fb.shape[1])
        res = subprocess.run([str(exe)], input=struct.pack("iii", fb.shape[0], fb.shape[1], 1154) + fb.tobytes() + s.tobytes(), capture_output=True)
        assert res.returncode == 0, res.returncode   # (3 / 4: a read outside the spectrum / a slot outside the partial sums)
        ok, steps, slots, max_parts = struct.unpack("4i", res.stdout[:16])
        assert ok == 1 and slots <= fb.shape[0] + 64 and steps * 64 < np.count_nonzero(fb) + 8 * 64
        mel = np.frombuffer(res.stdout[16:], dtype=np.float64)
        assert np.abs(mel - fb @ s).max() <= 4e-16 * np.abs(fb).sum(axis=1).max() * np.abs(s).max()
    # refused (the library then keeps the frame-per-workgroup kernel): partial sums that do not fit the LDS behind the spectrum, streams
    # longer than 256 steps
    for fb, spare in ((banks[0], 150), (rng.standard_normal((128, 1024)), 1154)):
        res = subprocess.run([str(exe)], input=struct.pack("iii", 128, 1024, spare) + fb.tobytes() + fb[0].tobytes(), capture_output=True)
        assert res.returncode == 0 and struct.unpack("4i", res.stdout[:16])[0] == 0


def test_run_sharded_with_a_fake_device_step():
    """The in-process sharder (threads + block partition) with a CPU stand-in for the device call."""
    calls = []

    def fake_batch(clips, scale, device=0):
        calls.append((device, clips.shape[0]))
        return clips * scale + device * 0.0

    x = np.arange(7 * 5, dtype=np.float32).reshape(7, 5)
    out = zafx.run_sharded(fake_batch, x, [0, 1, 2], 2.0)
    assert np.array_equal(out, x * 2.0)
    assert sorted(calls) == [(0, 2), (1, 2), (2, 3)]
    one = zafx.run_sharded(fake_batch, x[:1], [0, 1], 3.0)
    assert np.array_equal(one, x[:1] * 3.0)

    def failing(clips, device=0):
        raise RuntimeError("device lost")

    with pytest.raises(RuntimeError):
        zafx.run_sharded(failing, x, [0, 1])
    with pytest.raises(ValueError):
        zafx.run_sharded(fake_batch, x, [], 1.0)


def test_dct_dst_matrices_match_reference(golden):
    """Closed-form orthonormal DCT/DST matrices (the operands of the GPU GEMM) vs zaf.dct / zaf.dst."""
    g = golden["dctdst"]
    for n in (8, 9, 100, 1024):
        x = g[f"x_{n}"]
        for t in (1, 2, 3, 4):
            assert relerr(zafx.dct_matrix(n, t) @ x, g[f"dct{t}_{n}"]) <= 1e-11
            assert relerr(zafx.dst_matrix(n, t) @ x, g[f"dst{t}_{n}"]) <= 1e-11
            m = zafx.dct_matrix(n, t)
            assert np.max(np.abs(m @ m.T - np.eye(n))) < 1e-10   # orthonormal
    with pytest.raises(ValueError):
        zafx.dct_matrix(8, 5)
    with pytest.raises(ValueError):
        zafx.dct(np.zeros((2, 8)), 2)


def test_row_align_validation():
    """row_align is checked before any device call (the C-ABI repeats the check for other bindings)."""
    from zafx import core
    for bad in (3, 24, -2, 2048):
        with pytest.raises(ValueError):
            core._as_row_align(bad, "FT")
    with pytest.raises(ValueError):
        core._as_row_align(16, "TF")
    assert core._as_row_align(0, "TF") == 0 and core._as_row_align(None, "FT") == 0 and core._as_row_align(16, "FT") == 16


def test_window_length_routing_rules():
    """Which window lengths the host layer accepts, and which go to the float32 kernels (decided before any device call)."""
    from zafx import core
    assert [n for n in (32, 64, 100, 2048, 4096, 8192, 16384) if core._tuned(n)] == [64, 2048, 4096, 8192]
    # float32 kernels: tiled for the powers of two, Bluestein forms for the other lengths 33 ... 8192; float64 below that
    assert [n for n in (2, 16, 32, 33, 63, 64, 100, 1764, 2047, 2048, 2049, 4096, 8191, 8193) if core._f32_window(n)] == \
        [33, 63, 64, 100, 1764, 2047, 2048, 2049, 4096, 8191]
    for n in (2, 3, 63, 1000, 1764, 2047, 3000, 8191):          # any length up to 8192
        assert len(core._as_window(np.ones(n), any_length=True)) == n
    for n in (1, 8193, 16384):                                   # ... but not above it
        with pytest.raises(ValueError):
            core._as_window(np.ones(n), any_length=True)
    for n in (32, 1000, 3000):                                   # callers that need a float32 kernel
        with pytest.raises(ValueError):
            core._as_window(np.ones(n))
    with pytest.raises(ValueError):
        core._as_window(np.ones((4, 4)), any_length=True)
    for n in (999, 2):                                           # MDCT: even and >= 4, checked before the device
        with pytest.raises(ValueError):
            zafx.mdct_plan(np.ones(n))


def test_cqt_short_kernel_embedding():
    """core._cqt_embed: a kernel with fft_length below 512, rewritten for 512-sample frames, gives the numbers of the original
    (zaf.py:603-632) -- checked with the oracle's cqtspectrogram on both, no device involved."""
    from oracle import zaf_oracle as orc
    from zafx import core
    rng = np.random.default_rng(5)
    cases = [(orc.cqtkernel(16000, 2, 220.0, 7040.0), 16000, 100), (orc.cqtkernel(8000, 3, 440.0, 3520.0), 8000, 100),
             (scipy.sparse.csr_matrix(rng.standard_normal((7, 300)) + 1j * rng.standard_normal((7, 300))), 4040, 40)]   # any length
    for kern, fs, tr in cases:
        assert kern.shape[1] < 512
        big = core._cqt_embed(kern, round(fs / tr))
        assert big.shape == (kern.shape[0], 512)
        for n in (1, 37, 5000, 5001):
            x = rng.standard_normal(n)
            ref, got = orc.cqtspectrogram(x, fs, tr, kern), orc.cqtspectrogram(x, fs, tr, big)
            assert got.shape == ref.shape
            assert ref.size == 0 or relerr(got, ref) <= 1e-13


# ------------------------------------------------------------------------------------------------------
# device-buffer pool and plan cache with a stubbed library (no GPU): round-2 advisor findings
# ------------------------------------------------------------------------------------------------------
class _StubLib:
    """Stands in for libzafx.so: hands out fake device pointers and records what was freed."""

    def __init__(self):
        self.next, self.live, self.freed, self.fail_alloc_with = 0x1000, set(), [], 0

    def zafx_alloc(self, device, pptr, nbytes):
        if self.fail_alloc_with:
            rc, self.fail_alloc_with = self.fail_alloc_with, 0
            return rc
        self.next += 0x1000
        pptr._obj.value = self.next
        self.live.add(self.next)
        return 0

    def zafx_free(self, device, ptr):
        self.live.discard(ptr.value)
        self.freed.append(ptr.value)
        return 0

    def zafx_last_error(self):
        return b"stub"


@pytest.fixture
def stub_pool(monkeypatch):
    from zafx import _lib, core
    stub = _StubLib()
    monkeypatch.setattr(_lib, "load", lambda: stub)
    monkeypatch.setattr(core.DeviceBuffer, "_pool", {})
    monkeypatch.setattr(core.DeviceBuffer, "_pool_bytes", [0])
    monkeypatch.setattr(core.DeviceBuffer, "_POOL_CAP", 1000)
    return stub, core.DeviceBuffer


def test_pool_eviction_skips_size_classes_emptied_by_pooled(stub_pool):
    """pooled() popped the last pointer of the oldest size class and left its empty list at the head of the
    eviction order; the next release() that had to evict raised IndexError out of run_host's finally block."""
    stub, DB = stub_pool
    a = DB((400,), np.uint8)
    a.release()                      # oldest size class: 400 bytes parked
    b = DB((500,), np.uint8)
    b.release()                      # 900 of 1000 bytes parked
    got = DB.pooled((400,), np.uint8)   # takes the 400-byte pointer back: its class is now empty
    assert (0, 400) not in DB._pool and DB._pool_bytes[0] == 500
    c = DB((600,), np.uint8)
    c.release()                      # must evict the 500-byte class (500 + 600 > 1000), not trip over an empty list
    assert DB._pool_bytes[0] == 600 and list(DB._pool) == [(0, 600)]
    assert b.ptr.value is None or not b.ptr.value
    got.release()
    assert DB._pool_bytes[0] == 1000
    DB.drain_pool()
    assert not stub.live and DB._pool_bytes[0] == 0


def test_pool_is_drained_only_when_the_device_is_out_of_memory(stub_pool):
    stub, DB = stub_pool
    from zafx import _lib
    DB((300,), np.uint8).release()
    stub.fail_alloc_with = 101       # any other error (an invalid device ordinal, say): the pool stays
    with pytest.raises(zafx.ZafxError):
        DB((10,), np.uint8)
    assert DB._pool_bytes[0] == 300
    stub.fail_alloc_with = _lib.ERROR_OUT_OF_MEMORY
    keep = DB((10,), np.uint8)       # out of memory: parked allocations of that device are given back, one retry
    assert DB._pool_bytes[0] == 0 and len(stub.freed) == 1 and keep.ptr.value in stub.live


def test_plan_cache_builds_outside_its_lock(monkeypatch):
    """the factory of a missing plan ran under the cache lock (a 2 GB dct matrix stalled every other lookup)."""
    import threading
    from zafx import core
    monkeypatch.setattr(core, "_cache", {})
    seen = []

    class FakePlan:
        def __init__(self, tag):
            self.tag, self.destroyed = tag, False

        def destroy(self):
            self.destroyed = True

    def slow_factory():
        seen.append(core._cache_lock.acquire(blocking=False))   # the lock is free while a plan is being built
        if seen[-1]:
            core._cache_lock.release()
        return FakePlan("a")

    p = core._cached(("k",), slow_factory)
    assert seen == [True] and core._cached(("k",), lambda: FakePlan("b")) is p
    # two threads missing on one key: one plan survives, the other is destroyed
    gate, made = threading.Barrier(2), []

    def racing_factory():
        gate.wait(timeout=10)
        made.append(FakePlan("r"))
        return made[-1]

    out = []
    ts = [threading.Thread(target=lambda: out.append(core._cached(("race",), racing_factory))) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert out[0] is out[1] and len(made) == 2 and sorted(m.destroyed for m in made) == [False, True]


def test_f64_window_limits_are_checked_on_the_host():
    """f64=True with a window of 2049 ... 8191 samples that is not a power of two reached zafx_plan_create."""
    w = np.ones(3000)
    for call in (lambda: zafx.stft_plan(w, 500, f64=True), lambda: zafx.istft_plan(w, 500, f64=True),
                 lambda: zafx.mdct_plan(w, f64=True)):
        with pytest.raises(ValueError):
            call()
