"""Plans, device buffers and the batched / drop-in transform functions.

Layering: Python host code -> ctypes -> libzafx.so (C-ABI, include/zafx.h)
-> hand-written HIP kernels.  No PyTorch, no CPU fallback.
"""
import ctypes
import threading
import weakref

import numpy as np

from . import _lib
from . import constants
from ._lib import ZafxError

_LAYOUTS = {"FT": _lib.LAYOUT_FT, "TF": _lib.LAYOUT_TF, _lib.LAYOUT_FT: _lib.LAYOUT_FT, _lib.LAYOUT_TF: _lib.LAYOUT_TF}


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


# ======================================================================================
# device memory
# ======================================================================================
class _Pinned:
    """Owner of one page-locked host allocation (zafx_host_alloc); freed when the last array view dies."""

    def __init__(self, nbytes):
        p = ctypes.c_void_p()
        _lib.check(_lib.load().zafx_host_alloc(ctypes.byref(p), int(nbytes)), "zafx_host_alloc")
        self.ptr, self.nbytes = p, int(nbytes)

    def __del__(self):
        try:
            if self.ptr and self.ptr.value:
                _lib.load().zafx_host_free(self.ptr)
                self.ptr = ctypes.c_void_p()
        except Exception:
            pass


def pinned_empty(shape, dtype):
    """np.empty in page-locked host memory: transfers at the PCIe rate (~56 GB/s) instead of ~25 GB/s pageable."""
    dtype = np.dtype(dtype)
    shape = tuple(int(s) for s in np.atleast_1d(shape))
    nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
    if nbytes == 0:
        return np.empty(shape, dtype)
    owner = _Pinned(nbytes)
    raw = (ctypes.c_uint8 * nbytes).from_address(owner.ptr.value)
    arr = np.frombuffer(raw, dtype=dtype).reshape(shape)
    _PINNED_OWNERS[id(raw)] = owner   # keep the allocation alive as long as `raw` (the array's base) lives
    weakref.finalize(raw, _PINNED_OWNERS.pop, id(raw), None)
    return arr


_PINNED_OWNERS = {}


class DeviceBuffer:
    """A typed, shaped allocation in one GPU's HBM (owned by this object)."""

    def __init__(self, shape, dtype, device=0, _ptr_from_pool=None):
        self.shape = tuple(int(s) for s in np.atleast_1d(shape))
        self.dtype = np.dtype(dtype)
        self.device = int(device)
        self.nbytes = int(np.prod(self.shape, dtype=np.int64)) * self.dtype.itemsize
        if _ptr_from_pool is not None:
            self.ptr = _ptr_from_pool
            return
        p = ctypes.c_void_p()
        rc = _lib.load().zafx_alloc(self.device, ctypes.byref(p), self.nbytes)
        if rc == _lib.ERROR_OUT_OF_MEMORY and DeviceBuffer._pool_bytes[0] > 0:
            # out of device memory with allocations parked in the pool: give back those of THIS device, retry once
            # (any other error -- a bad device index, say -- leaves the pool alone)
            DeviceBuffer.drain_pool(self.device)
            rc = _lib.load().zafx_alloc(self.device, ctypes.byref(p), self.nbytes)
        _lib.check(rc, "zafx_alloc")
        self.ptr = p

    # Allocation pool of the host-buffer entry points: a drop-in call on one 10 s clip spends more time in
    # hipMalloc / hipFree (which synchronises the device) than in its kernel, so run_host recycles allocations
    # of the exact size it needs (repeated calls have repeated sizes), up to _POOL_CAP bytes per process.
    _pool, _pool_bytes, _pool_lock = {}, [0], threading.Lock()
    _POOL_CAP = 4 << 30

    @classmethod
    def pooled(cls, shape, dtype, device=0):
        probe_bytes = int(np.prod([int(s) for s in np.atleast_1d(shape)], dtype=np.int64)) * np.dtype(dtype).itemsize
        with cls._pool_lock:
            key = (int(device), probe_bytes)
            free = cls._pool.get(key)
            ptr = free.pop() if free else None
            if ptr is not None:
                cls._pool_bytes[0] -= probe_bytes
            if free is not None and not free:
                del cls._pool[key]   # (an empty size class must not sit at the head of the eviction order, see release)
        return cls(shape, dtype, device, _ptr_from_pool=ptr) if ptr is not None else cls(shape, dtype, device)

    def release(self):
        """Return the allocation to the pool.  The pool is bounded (_POOL_CAP bytes): the oldest parked allocations are
        freed to make room, so a workload of ever-changing clip lengths cannot pin device memory with sizes it never reuses."""
        if getattr(self, "ptr", None) is None or not self.ptr.value:
            return
        if self.nbytes == 0 or self.nbytes > self._POOL_CAP:
            self.free()
            return
        evicted = []
        with self._pool_lock:
            while self._pool_bytes[0] + self.nbytes > self._POOL_CAP and self._pool:
                key = next(iter(self._pool))            # dicts keep insertion order: the size class parked first
                ptrs = self._pool[key]
                if not ptrs:                            # (never left behind by pooled(); kept as a guard)
                    del self._pool[key]
                    continue
                evicted.append((key[0], ptrs.pop(0)))
                self._pool_bytes[0] -= key[1]
                if not ptrs:
                    del self._pool[key]
            self._pool.setdefault((self.device, self.nbytes), []).append(self.ptr)
            self._pool_bytes[0] += self.nbytes
        self.ptr = ctypes.c_void_p()
        for device, ptr in evicted:
            _lib.load().zafx_free(device, ptr)

    @classmethod
    def drain_pool(cls, device=None):
        """Free the parked allocations (of one device, or of all)."""
        with cls._pool_lock:
            items = {k: v for k, v in cls._pool.items() if device is None or k[0] == int(device)}
            for k, ptrs in items.items():
                del cls._pool[k]
                cls._pool_bytes[0] -= k[1] * len(ptrs)
        for (dev, _), ptrs in items.items():
            for p in ptrs:
                _lib.load().zafx_free(dev, p)

    @classmethod
    def from_host(cls, array, device=0):
        array = np.ascontiguousarray(array)
        buf = cls(array.shape, array.dtype, device)
        buf.upload(array)
        return buf

    def upload(self, array):
        array = np.ascontiguousarray(array, dtype=self.dtype)
        if array.nbytes != self.nbytes:
            raise ValueError("upload size mismatch")
        if self.nbytes:
            _lib.check(_lib.load().zafx_h2d(self.device, self.ptr, _ptr(array), self.nbytes), "zafx_h2d")
        return self

    def download(self, first=0, count=None, out=None):
        """Copy to host; `first`/`count` select a range along axis 0.  `out`: destination array (e.g. from
        pinned_empty, reused across calls -- pinning costs ~0.2 ms/MB, so it only pays for a buffer that lives on)."""
        count = self.shape[0] - first if count is None else count
        if first < 0 or count < 0 or first + count > self.shape[0]:
            raise ValueError("download range out of bounds")
        shape = (count,) + self.shape[1:]
        if out is None:
            out = np.empty(shape, dtype=self.dtype)
        elif out.shape != shape or out.dtype != self.dtype or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous array of the downloaded shape and dtype")
        if out.nbytes:
            row = self.nbytes // max(self.shape[0], 1)
            src = ctypes.c_void_p(self.ptr.value + first * row)
            _lib.check(_lib.load().zafx_d2h(self.device, _ptr(out), src, out.nbytes), "zafx_d2h")
        return out

    def fill_zero(self):
        if self.nbytes:
            _lib.check(_lib.load().zafx_memset(self.device, self.ptr, 0, self.nbytes), "zafx_memset")
        return self

    def copy_from(self, other, nbytes=None, dst_offset=0, src_offset=0):
        nbytes = other.nbytes if nbytes is None else nbytes
        dst = ctypes.c_void_p(self.ptr.value + dst_offset)
        src = ctypes.c_void_p(other.ptr.value + src_offset)
        _lib.check(_lib.load().zafx_d2d(self.device, dst, src, nbytes), "zafx_d2d")
        return self

    @classmethod
    def placed(cls, shape, dtype, probe, candidates=4, device=0, init=None):
        """The fastest of `candidates` allocations of this shape: -> (buffer, [probe times]).

        Where a large allocation lands in physical memory changes the rate of the kernels that walk it with a row stride (the
        (W, T) spectra of the reference layout): the same STFT runs in 1.50 ms into one 7 GB allocation and in 1.70 ms into
        another made by the same process (profiles/r02_notes.md).  The address is not the caller's to choose, so a long-lived
        buffer is picked by trial: all candidates are allocated (held at once -- freed ones would come straight back), `init(buffer)`
        fills each if the probe reads it, `probe(buffer)` returns a time, the best one stays, the others are freed."""
        bufs = [cls(shape, dtype, device) for _ in range(max(int(candidates), 1))]
        times = []
        for k, b in enumerate(bufs):
            if init is not None:
                init(b)
            if k == 0 and len(bufs) > 1:
                probe(b)   # (not counted: the first probe also carries the clock ramp)
            times.append(float(probe(b)))
        best = min(range(len(bufs)), key=times.__getitem__)
        for i, b in enumerate(bufs):
            if i != best:
                b.free()
        return bufs[best], times

    @classmethod
    def placed_for(cls, plan, d_in, n_clips, n_in, candidates=4, reps=8):
        """The output buffer of `plan` for (n_clips, n_in) as the fastest of `candidates` allocations, picked by the library itself
        (zafx_alloc_placed: every candidate timed with the plan's own kernel on `d_in`): -> (buffer, [ms per launch of each candidate]).
        The C-ABI twin of `placed` -- what a caller without this Python layer uses."""
        shape, dtype = plan.out_shape(n_clips, n_in), plan.out_dtype
        nbytes = int(np.prod(shape, dtype=np.int64)) * dtype.itemsize
        p = ctypes.c_void_p()
        times = (ctypes.c_float * int(candidates))()
        _lib.check(_lib.load().zafx_alloc_placed(plan.handle, ctypes.byref(p), nbytes, d_in.ptr, int(n_clips), int(n_in), int(candidates), int(reps), times),
                   "zafx_alloc_placed")
        return cls(shape, dtype, plan.device, _ptr_from_pool=p), [float(t) for t in times]

    def free(self):
        if getattr(self, "ptr", None) is not None and self.ptr.value:
            _lib.load().zafx_free(self.device, self.ptr)
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ======================================================================================
# plans
# ======================================================================================
# `onesided` of the STFT / ISTFT entry points: False / True, or "magnitude" / "power" (STFT only: |X| or |X|^2 of
# rows 0..W/2 as real arrays -- the spectrogram of the reference's examples, zaf.py:83)
_SPECTRA = {False: _lib.SPECTRUM_TWO_SIDED, True: _lib.SPECTRUM_ONE_SIDED, "magnitude": _lib.SPECTRUM_MAGNITUDE,
            "power": _lib.SPECTRUM_POWER}


def _as_row_align(row_align, layout):
    a = int(row_align or 0)
    if a < 0 or a > 1024 or a & (a - 1):
        raise ValueError("row_align must be 0 or a power of two <= 1024")
    if a > 1 and _LAYOUTS[layout] != _lib.LAYOUT_FT:
        raise ValueError('row_align applies to layout "FT" only')
    return a


def _spectrum_of(onesided):
    try:
        return _SPECTRA[onesided]
    except (KeyError, TypeError):
        raise ValueError('onesided must be False, True, "magnitude" or "power"') from None


def _rows_with_pitch(a, pitch):
    """a: (B, F, T).  When `a` is a view of rows that lie `pitch` elements apart inside one buffer -- clip after clip, as the (B, F, pitch)
    array of a row_align plan does -- return that (B, F, pitch) array over the same memory, else None.  (The padding behind the last
    row must belong to the buffer too: checked against the bounds of the array that owns the memory.)"""
    isz = a.dtype.itemsize
    b, f, t = a.shape
    if t > pitch or a.strides != (f * pitch * isz, pitch * isz, isz) or not a.size:
        return None
    owner = a
    while isinstance(owner.base, np.ndarray):
        owner = owner.base
    try:
        from numpy.lib.array_utils import byte_bounds
    except ImportError:   # NumPy < 2
        byte_bounds = np.byte_bounds
    lo, hi = byte_bounds(owner)
    start = a.__array_interface__["data"][0]
    if start < lo or start + b * f * pitch * isz > hi:
        return None
    return np.lib.stride_tricks.as_strided(a, shape=(b, f, pitch), strides=(f * pitch * isz, pitch * isz, isz), writeable=False)


def _line_elements(dtype):
    """Elements of one 128-byte line."""
    return 128 // np.dtype(dtype).itemsize


class Plan:
    """One transform kind bound to one device and one HIP stream (zafx_plan)."""

    _FORWARD = (_lib.STFT, _lib.MDCT, _lib.MEL, _lib.MFCC, _lib.CQT, _lib.CHROMA)   # 2-D (frequency x time) outputs

    def __init__(self, kind, device=0, window_length=0, step_length=0, layout="FT", n_filters=0, n_coefs=0,
                 fft_length=0, n_bins=0, octave_resolution=0, onesided=False, f64=False, row_align=0, transform_type=0, transform_sine=False,
                 with_mel=False):
        self.kind = kind
        self.device = int(device)
        self.layout = _LAYOUTS[layout]
        prm = _lib.ZafxParams()
        prm.struct_size = ctypes.sizeof(_lib.ZafxParams)
        prm.window_length = int(window_length)
        prm.step_length = int(step_length)
        prm.layout = self.layout
        prm.n_filters = int(n_filters)
        prm.n_coefs = int(n_coefs)
        prm.fft_length = int(fft_length)
        prm.n_bins = int(n_bins)
        prm.octave_resolution = int(octave_resolution)
        prm.transform_type = int(transform_type)
        prm.transform_sine = int(bool(transform_sine))
        prm.with_mel = int(bool(with_mel))
        self.spectrum = _spectrum_of(onesided)
        prm.spectrum = self.spectrum
        prm.precision = _lib.PRECISION_F64 if f64 else _lib.PRECISION_F32
        self.f64 = bool(f64)
        self.row_align = _as_row_align(row_align, layout)
        prm.row_align = self.row_align
        self.params = prm
        h = ctypes.c_void_p()
        _lib.check(_lib.load().zafx_plan_create(ctypes.byref(h), self.device, kind, ctypes.byref(prm)), "zafx_plan_create")
        self.handle = h
        self.lock = threading.Lock()

    # ---- constants -----------------------------------------------------------------
    def _set(self, which, array, dtype):
        array = np.ascontiguousarray(array, dtype=dtype)
        _lib.check(_lib.load().zafx_plan_set_constant(self.handle, which, _ptr(array), array.nbytes), "zafx_plan_set_constant")

    def set_window(self, window_function):
        self._set(_lib.CONST_WINDOW, window_function, np.float64 if self.f64 else np.float32)

    def set_mel_filterbank(self, mel_filterbank):
        dense = mel_filterbank.toarray() if hasattr(mel_filterbank, "toarray") else np.asarray(mel_filterbank)
        self._set(_lib.CONST_MEL_FB, dense, np.float64 if self.f64 else np.float32)

    def set_dct(self, dct_rows):
        self._set(_lib.CONST_DCT, dct_rows, np.float64 if self.f64 else np.float32)

    def set_matrix(self, matrix):
        self._set(_lib.CONST_MATRIX, matrix, np.float32)

    def set_cqt_kernel(self, cqt_kernel):
        csr = cqt_kernel.tocsr()
        csr.sort_indices()
        self._set(_lib.CONST_CQT_INDPTR, csr.indptr, np.int32)
        self._set(_lib.CONST_CQT_INDICES, csr.indices, np.int32)
        self._set(_lib.CONST_CQT_VALUES, csr.data, np.complex128 if self.f64 else np.complex64)

    # ---- geometry / execution ----------------------------------------------------------
    def out_dims(self, n_in):
        dims = (ctypes.c_int64 * 2)()
        _lib.check(_lib.load().zafx_plan_out_dims(self.handle, int(n_in), dims), "zafx_plan_out_dims")
        return int(dims[0]), int(dims[1])

    def row_pitch(self, n_in):
        """Elements between consecutive rows of the plan's 2-D array (T rounded up to row_align; T for compact plans)."""
        pitch = ctypes.c_int64()
        _lib.check(_lib.load().zafx_plan_row_pitch(self.handle, int(n_in), ctypes.byref(pitch)), "zafx_plan_row_pitch")
        return int(pitch.value)

    def out_shape(self, n_clips, n_in):
        """Shape of the device array execute() writes; with row_align the last axis of an (F, T) array is the pitch."""
        rows, frames = self.out_dims(n_in)
        if self.kind in self._FORWARD:
            return (n_clips, rows, self.row_pitch(n_in)) if self.layout == _lib.LAYOUT_FT else (n_clips, frames, rows)
        return (n_clips, rows)

    @property
    def out_dtype(self):
        complex_out = self.kind == _lib.STFT and self.spectrum < _lib.SPECTRUM_MAGNITUDE
        if self.f64:
            return np.dtype(np.complex128) if complex_out else np.dtype(np.float64)
        return np.dtype(np.complex64) if complex_out else np.dtype(np.float32)

    @property
    def in_dtype(self):
        if self.kind == _lib.ISTFT:
            return np.dtype(np.complex128) if self.f64 else np.dtype(np.complex64)
        return np.dtype(np.float64) if self.f64 else np.dtype(np.float32)

    def execute(self, d_in, d_out, n_clips, n_in):
        """Enqueue on the plan's stream (asynchronous); d_in / d_out are DeviceBuffers."""
        _lib.check(_lib.load().zafx_execute(self.handle, d_in.ptr, d_out.ptr, int(n_clips), int(n_in)), "zafx_execute")

    def sync(self):
        _lib.check(_lib.load().zafx_sync(self.handle), "zafx_sync")

    def pcm_to_float(self, d_pcm, d_out, n_clips, n_frames, n_channels):
        """Enqueue wavread's normalisation + channel mean on this plan's stream (int16/int32 PCM in)."""
        _lib.check(_lib.load().zafx_pcm_to_float(self.handle, d_pcm.ptr, d_out.ptr, int(n_clips), int(n_frames),
                                                int(n_channels), d_pcm.dtype.itemsize), "zafx_pcm_to_float")

    def execute_pcm(self, d_pcm, d_out, n_clips, n_frames, n_channels=1):
        """execute() on integer PCM that is already on the device: d_pcm = (clips, frames[, channels]) int16 / int32 interleaved.  At window 2048
        mel, mfcc (and their one-pass form), the complex and the |X| / |X|^2 kinds of the STFT and the MDCT read int16 (one or two channels)
        in their own loads -- 2 bytes per sample and channel of HBM traffic instead of the pre-pass's 6 + 4 --, every other plan converts,
        in chunks of clips, into a bounded staging array it owns first (zafx_execute_pcm, include/zafx.h; zaf.py:1202 and :65 either way)."""
        if d_pcm.dtype not in (np.dtype(np.int16), np.dtype(np.int32)):
            raise ValueError("execute_pcm takes an int16 or int32 DeviceBuffer (wavread's other dtypes: convert on the host)")
        n_clips, n_frames, n_channels = int(n_clips), int(n_frames), int(n_channels)
        if int(np.prod(d_pcm.shape, dtype=np.int64)) < n_clips * n_frames * n_channels:
            raise ValueError("d_pcm holds fewer than clips x frames x channels samples")
        if d_out.nbytes < n_clips * self.clip_bytes(n_frames)[1]:
            raise ValueError("d_out is smaller than the plan's output for these clips")
        with self.lock:   # (the plan's staging array of the two-step route is shared state)
            _lib.check(_lib.load().zafx_execute_pcm(self.handle, d_pcm.ptr, d_out.ptr, n_clips, n_frames, n_channels,
                                                   d_pcm.dtype.itemsize), "zafx_execute_pcm")

    def timer_start(self):
        _lib.check(_lib.load().zafx_timer_start(self.handle), "zafx_timer_start")

    def timer_stop(self):
        ms = ctypes.c_float(0)
        _lib.check(_lib.load().zafx_timer_stop(self.handle, ctypes.byref(ms)), "zafx_timer_stop")
        return ms.value

    @property
    def kernel_name(self):
        """The kernel family the plan was built for (does not change with the calls)."""
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().zafx_plan_kernel_name(self.handle, buf, 128), "zafx_plan_kernel_name")
        return buf.value.decode()

    @property
    def last_kernel(self):
        """The kernel the last execute / run_host really launched (carry, band and generic forms are chosen per call);
        the planned name before the first one."""
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().zafx_plan_last_kernel_name(self.handle, buf, 128), "zafx_plan_last_kernel_name")
        return buf.value.decode() or self.kernel_name

    def clip_bytes(self, n_in):
        """(input bytes, output bytes) of ONE clip for `n_in` (zafx_plan_clip_bytes; rows at the plan's pitch)."""
        a, b = ctypes.c_int64(), ctypes.c_int64()
        _lib.check(_lib.load().zafx_plan_clip_bytes(self.handle, int(n_in), ctypes.byref(a), ctypes.byref(b)), "zafx_plan_clip_bytes")
        return int(a.value), int(b.value)

    def run_host(self, array, n_in, out=None, chunk_clips=0):
        """Host array in -> device transform -> host array out: the reference's own boundary (zaf.py:45), PCIe both ways.

        One call of zafx_run_host: the clips travel in chunks over two HIP streams with plan-owned staging buffers in HBM,
        so the upload, the kernel and the download of neighbouring chunks overlap.  `out`: destination array of
        Plan.out_shape(n_clips, n_in) and Plan.out_dtype -- with page-locked arrays on both sides (pinned_empty, reused
        across calls) the call runs at the PCIe rate of its slower direction; pageable arrays work and are staged."""
        array = np.asarray(array)
        if np.iscomplexobj(array) and self.in_dtype.kind != "c":
            raise ValueError("this plan takes real input")
        frames = None
        if self.row_align > 1 and self.kind not in self._FORWARD and array.ndim == 3:
            # padded rows on the device, compact (F, T) indexing on the host side.  An array that already IS rows of this pitch -- the view a
            # forward *_batch call handed back -- goes as it lies; anything else is copied into a padded array first
            pitch = self.row_pitch(n_in)
            whole = _rows_with_pitch(array, pitch) if array.dtype == self.in_dtype else None
            if whole is None:
                whole = np.zeros(array.shape[:2] + (pitch,), dtype=self.in_dtype)
                whole[:, :, :array.shape[2]] = array
            array = whole
        array = np.ascontiguousarray(array, dtype=self.in_dtype)   # (an array of another dtype would be reinterpreted byte-wise)
        n_clips = array.shape[0]
        if self.row_align > 1 and self.kind in self._FORWARD:
            frames = self.out_dims(n_in)[1]
        shape = self.out_shape(n_clips, n_in)
        if out is None:
            out = np.empty(shape, dtype=self.out_dtype)
        elif tuple(out.shape) != tuple(shape) or out.dtype != self.out_dtype or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError(f"out must be a writeable C-contiguous {self.out_dtype} array of shape {tuple(shape)}")
        in_b, out_b = self.clip_bytes(n_in)
        if array.nbytes != n_clips * in_b or out.nbytes != n_clips * out_b:   # (the library walks both arrays by these sizes)
            raise ValueError(f"array sizes do not match the plan: {array.nbytes} B in for {n_clips} clips of {in_b} B, "
                             f"{out.nbytes} B out for clips of {out_b} B")
        if n_clips and out.nbytes:
            with self.lock:
                _lib.check(_lib.load().zafx_run_host(self.handle, _ptr(array), _ptr(out), n_clips, int(n_in), int(chunk_clips)),
                           "zafx_run_host")
        return out if frames is None else out[:, :, :frames]

    def run_host_pcm(self, pcm, out=None, chunk_clips=0):
        """run_host for integer PCM as wavread's source holds it (zaf.py:1187-1204): pcm = (clips, frames[, channels]) int16 or
        int32, interleaved.  The integers cross PCIe (2-4 bytes per sample and channel), x / 2^(bits-1) (zaf.py:1202) and the
        channel mean (zaf.py:65) run on the device in front of the transform (zafx_run_host_pcm)."""
        pcm = np.ascontiguousarray(pcm)
        if pcm.dtype not in (np.dtype(np.int16), np.dtype(np.int32)):
            raise ValueError("PCM ingest takes int16 or int32 samples (wavread's other dtypes: convert on the host)")
        if pcm.ndim == 2:
            pcm = pcm[:, :, None]
        if pcm.ndim != 3:
            raise ValueError("pcm must be (clips, frames) or (clips, frames, channels)")
        if self.f64 or self.kind not in self._FORWARD + (_lib.DCT,):
            raise ValueError("PCM ingest feeds the float32 plans that take samples")
        n_clips, n_in, channels = pcm.shape
        frames = self.out_dims(n_in)[1] if self.row_align > 1 else None
        shape = self.out_shape(n_clips, n_in)
        if out is None:
            out = np.empty(shape, dtype=self.out_dtype)
        elif tuple(out.shape) != tuple(shape) or out.dtype != self.out_dtype or not out.flags.c_contiguous or not out.flags.writeable:
            raise ValueError(f"out must be a writeable C-contiguous {self.out_dtype} array of shape {tuple(shape)}")
        if n_clips and out.nbytes:
            with self.lock:
                _lib.check(_lib.load().zafx_run_host_pcm(self.handle, _ptr(pcm), _ptr(out), n_clips, n_in, channels, pcm.dtype.itemsize,
                                                         int(chunk_clips)), "zafx_run_host_pcm")
        return out if frames is None else out[:, :, :frames]

    def destroy(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            _lib.load().zafx_plan_destroy(self.handle)
            self.handle = ctypes.c_void_p()

    def __del__(self):
        try:
            self.destroy()
        except Exception:
            pass


class Comm:
    """RCCL communicator (one process per GPU); only used to broadcast plan constants."""

    @staticmethod
    def unique_id():
        buf = ctypes.create_string_buffer(128)
        _lib.check(_lib.load().zafx_comm_unique_id(buf), "zafx_comm_unique_id")
        return buf.raw

    def __init__(self, device, rank, n_ranks, unique_id):
        if len(unique_id) != 128:
            raise ValueError("unique_id must be 128 bytes")
        self._id = ctypes.create_string_buffer(bytes(unique_id), 128)
        h = ctypes.c_void_p()
        _lib.check(_lib.load().zafx_comm_create(ctypes.byref(h), int(device), int(rank), int(n_ranks), self._id), "zafx_comm_create")
        self.handle = h
        self.rank, self.n_ranks = int(rank), int(n_ranks)

    def count(self):
        """Number of ranks in the communicator as RCCL reports it (ncclCommCount)."""
        n = ctypes.c_int(0)
        _lib.check(_lib.load().zafx_comm_count(self.handle, ctypes.byref(n)), "zafx_comm_count")
        return n.value

    def user_rank(self):
        """This process's rank as RCCL reports it (ncclCommUserRank)."""
        n = ctypes.c_int(-1)
        _lib.check(_lib.load().zafx_comm_user_rank(self.handle, ctypes.byref(n)), "zafx_comm_user_rank")
        return n.value

    def broadcast_constants(self, plan, root=0):
        _lib.check(_lib.load().zafx_comm_broadcast_constants(self.handle, plan.handle, int(root)), "zafx_comm_broadcast_constants")

    def destroy(self):
        if getattr(self, "handle", None) is not None and self.handle.value:
            _lib.load().zafx_comm_destroy(self.handle)
            self.handle = ctypes.c_void_p()


# ======================================================================================
# plan cache (thread safe): plans are keyed by geometry + a digest of their constants
# ======================================================================================
_cache = {}
_cache_lock = threading.Lock()
_CACHE_MAX = 32


def _digest(*arrays):
    import hashlib
    h = hashlib.blake2b(digest_size=16)
    for a in arrays:
        a = np.ascontiguousarray(a)
        h.update(str(a.dtype).encode())
        h.update(str(a.shape).encode())
        h.update(a.tobytes())
    return h.hexdigest()


def _cached(key, factory):
    """The plan under `key`, built by `factory()` on a miss.  The factory runs OUTSIDE the cache lock (a dct / dst plan builds
    an N x N float64 matrix, O(N^2) trigonometry: other threads' lookups must not wait for it); when two threads miss on the
    same key at once both build, the first insert wins and the loser's plan is destroyed."""
    with _cache_lock:
        plan = _cache.get(key)
    if plan is not None:
        return plan
    fresh = factory()
    with _cache_lock:
        plan = _cache.get(key)
        if plan is None:
            if len(_cache) >= _CACHE_MAX:
                _cache.pop(next(iter(_cache)))   # freed by Plan.__del__ once no caller still holds it
            _cache[key] = plan = fresh
            fresh = None
    if fresh is not None:
        fresh.destroy()
    return plan


def clear_plan_cache():
    with _cache_lock:
        for plan in _cache.values():
            plan.destroy()
        _cache.clear()
    DeviceBuffer.drain_pool()


# ======================================================================================
# argument checking shared by the drop-in and batched entry points
# ======================================================================================
def _pow2(n):
    return n > 0 and not n & (n - 1)


def _tuned(n):
    """Window lengths of the tiled float32 kernels (powers of two)."""
    return _pow2(n) and 64 <= n <= 8192


def _f32_window(n):
    """Window lengths with float32 kernels: the powers of two 64 ... 8192 (tiled kernels) and every other length 33 ... 8192
    (float32 Bluestein forms of STFT / ISTFT / MDCT / IMDCT, zafx_bs32.hip); the rest (below 33 samples) runs in float64."""
    return _tuned(n) or (33 <= n <= 8192 and not _pow2(n))


def _as_window(window_function, any_length=False):
    """any_length: windows that are not a power of two, up to 8192 samples (float32 Bluestein forms; the float64 ones stop
    at 2048)."""
    w = np.asarray(window_function, dtype=np.float64)
    if w.ndim != 1:
        raise ValueError("window_function must be 1-D")
    n = len(w)
    if any_length and not _tuned(n):
        if n < 2 or n > 8192:
            raise ValueError(f"zafx takes windows of 2 ... 8192 samples, got {n}")
        return w
    if not _tuned(n):
        raise ValueError(f"zafx kernels need a power-of-two window_length in [64, 8192], got {n}")
    return w


def _check_f64_window(n, f64):
    """float64 plans take the powers of two 64 ... 8192 and any other length up to 2048 (the float64 Bluestein forms stop
    there; the float32 ones reach 8192): validated here, ahead of the device."""
    if f64 and not _pow2(n) and n > 2048:
        raise ValueError(f"f64=True takes windows that are not a power of two only up to 2048 samples, got {n}")


def _as_step(step_length):
    if isinstance(step_length, (bool, np.bool_)) or not isinstance(step_length, (int, np.integer)):
        raise ValueError("step_length must be an int (as in zaf.stft)")
    if step_length < 1:
        raise ValueError("step_length must be >= 1")
    return int(step_length)


def _as_clips(audio, ndim_name="audio_signal", dtype=np.float32):
    a = np.asarray(audio)
    if a.ndim != 2:
        raise ValueError(f"{ndim_name} batch must be 2-D (clips, samples)")
    if np.iscomplexobj(a):
        raise ValueError(f"{ndim_name} must be real")
    return np.ascontiguousarray(a, dtype=dtype)


def _as_signal(audio_signal, dtype=np.float32):
    a = np.asarray(audio_signal)
    if a.ndim != 1:
        raise ValueError("audio_signal must be 1-D (one clip); use the *_batch functions for (clips, samples)")
    return _as_clips(a[None, :], dtype=dtype)


# Arithmetic of the drop-in transforms (every zaf.* function on the path): "f32" (default, the tuned kernels) or "f64" (the
# reference's own dtype on the device, within 1e-12 of zaf.py).
_PRECISION = {"value": "f32"}


def set_precision(precision):
    """Select the device arithmetic of the drop-in transforms (`stft`, `istft`, `mdct`, `imdct`, `melspectrogram`, `mfcc`,
    `cqtspectrogram`, `cqtchromagram`): "f32" (the tuned kernels) or "f64" (the reference's own dtype, within 1e-12 of it)."""
    if precision not in ("f32", "f64"):
        raise ValueError('precision must be "f32" or "f64"')
    _PRECISION["value"] = precision


def get_precision():
    return _PRECISION["value"]


# ======================================================================================
# plan factories
# ======================================================================================
def stft_plan(window_function, step_length, layout="FT", device=0, onesided=False, f64=False, row_align=0):
    """row_align (every 2-D plan factory): pad the rows of the device (F, T) array to a multiple of this many elements
    (16 for complex64, 32 for float32 = one 128-byte line) so that the reference-layout kernels run at their aligned
    rate for any T; Plan.out_shape / Plan.row_pitch give the padded geometry, 0 keeps the reference's compact order.
    A window outside the float32 kernels (below 33 samples) yields a float64 plan whatever `f64` says: Plan.in_dtype /
    Plan.out_dtype tell which arrays it takes."""
    w, h = _as_window(window_function, any_length=True), _as_step(step_length)   # (a hop above the window skips samples, as zaf.stft does)
    f64 = bool(f64) or not _f32_window(len(w))   # (windows below 33 samples: float64 Bluestein kernels)
    _check_f64_window(len(w), f64)
    key = ("stft", device, len(w), h, _LAYOUTS[layout], _spectrum_of(onesided), bool(f64), _as_row_align(row_align, layout), _digest(w))

    def make():
        p = Plan(_lib.STFT, device, window_length=len(w), step_length=h, layout=layout, onesided=onesided, f64=f64,
                 row_align=row_align)
        p.set_window(w)
        return p
    return _cached(key, make)


def istft_plan(window_function, step_length, layout="FT", device=0, onesided=False, f64=False, row_align=0):
    w, h = _as_window(window_function, any_length=True), _as_step(step_length)
    if h > len(w):
        raise ValueError("step_length must not exceed window_length")
    if onesided not in (False, True):
        raise ValueError("istft takes a complex spectrum: onesided must be False or True")
    # the tiled float32 overlap-add keeps 16 frames in LDS (8 at W = 4096, 4 at 8192); for a hop so small that more frames than
    # that cover one sample -- and for every window that is not a power of two -- the library takes the float32 frames +
    # gather overlap-add form of zafx_bs32.hip, which has no such limit (zafx_plan_create decides)
    f64 = bool(f64) or not _f32_window(len(w))
    _check_f64_window(len(w), f64)
    key = ("istft", device, len(w), h, _LAYOUTS[layout], bool(onesided), bool(f64), _as_row_align(row_align, layout), _digest(w))

    def make():
        p = Plan(_lib.ISTFT, device, window_length=len(w), step_length=h, layout=layout, onesided=onesided, f64=f64,
                 row_align=row_align)
        p.set_window(w)
        return p
    return _cached(key, make)


def mdct_plan(window_function, layout="FT", device=0, inverse=False, row_align=0, f64=False):
    w = _as_window(window_function, any_length=True)
    if len(w) % 2 or len(w) < 4:
        raise ValueError("the MDCT needs an even window_length >= 4")
    f64 = bool(f64) or not _f32_window(len(w))   # (even lengths below 34: float64 Bluestein kernels)
    _check_f64_window(len(w), f64)
    key = ("imdct" if inverse else "mdct", device, len(w), _LAYOUTS[layout], _as_row_align(row_align, layout), bool(f64), _digest(w))

    def make():
        p = Plan(_lib.IMDCT if inverse else _lib.MDCT, device, window_length=len(w), layout=layout, row_align=row_align, f64=f64)
        p.set_window(w)
        return p
    return _cached(key, make)


def _dense_filterbank(mel_filterbank, window_length):
    if not hasattr(mel_filterbank, "toarray"):
        raise ValueError("mel_filterbank must be a scipy.sparse matrix (as returned by melfilterbank)")
    fb = np.asarray(mel_filterbank.toarray(), dtype=np.float64)
    if fb.ndim != 2 or fb.shape[1] != window_length // 2:
        raise ValueError("mel_filterbank must have window_length/2 columns")
    return fb


def mel_plan(window_function, step_length, mel_filterbank, number_coefficients=None, layout="FT", device=0, row_align=0, f64=False, also_mel=False):
    """also_mel (with number_coefficients): the one-pass melspectrogram + mfcc plan (zafx_params.with_mel) -- its output holds the melspectrogram's
    rows, then the MFCCs' (mel_mfcc_batch splits them).  float32, window 2048, up to 128 filters and 32 coefficients: mel_mfcc_supported()."""
    w, h = _as_window(window_function, any_length=True), _as_step(step_length)
    if not hasattr(mel_filterbank, "toarray"):
        raise ValueError("mel_filterbank must be a scipy.sparse matrix (as returned by melfilterbank)")
    if mel_filterbank.ndim != 2 or mel_filterbank.shape[1] != len(w) // 2:
        raise ValueError("mel_filterbank must have window_length/2 columns")
    n_filters = mel_filterbank.shape[0]
    mfcc = number_coefficients is not None
    ncoef = int(number_coefficients) if mfcc else 0
    if mfcc and not 1 <= ncoef <= n_filters - 1:
        raise ValueError("number_coefficients must be in [1, number_filters - 1]")
    if len(w) > 8192:
        raise ValueError(f"melspectrogram / mfcc take windows of up to 8192 samples, got {len(w)}")
    # W = 4096: the fused two-band kernel; W = 8192 and windows that are not a power of two (33 ... 8192 samples: the float32 Bluestein
    # STFT) run as a spectrum kernel + the banded filterbank kernel k_melfb over a plan-owned scratch, in float32 -- as do filterbanks of 257 ... 576 rows at any
    # window; filterbanks above 576 rows and windows below 33 samples run on the float64 kernel (any power-of-two window and, up to 2048 samples, any other length)
    f64 = bool(f64) or n_filters > 576 or not _f32_window(len(w))
    if f64 and len(w) > 2048 and not _pow2(len(w)):
        raise ValueError(f"melspectrogram / mfcc in float64 (f64=True, or more than 576 filters) take windows of up to 2048 samples or the powers of two 4096 and 8192, got {len(w)}")
    # the cache key hashes the sparse triplet (a few KB), not the dense matrix (1 MB: 1.8 ms per call)
    csr = mel_filterbank.tocsr()
    also_mel = bool(also_mel)
    if also_mel and not mel_mfcc_supported(len(w), n_filters, ncoef if mfcc else 0, f64):
        raise ValueError("also_mel needs number_coefficients <= 32, a float32 plan of window 2048 and up to 128 filters (mel_mfcc_batch runs two plans otherwise)")
    key = ("mfcc" if mfcc else "mel", device, len(w), h, _LAYOUTS[layout], ncoef, n_filters, _as_row_align(row_align, layout), bool(f64), also_mel,
           _digest(w, csr.data, csr.indices, csr.indptr))

    def make():
        fb = _dense_filterbank(mel_filterbank, len(w))
        p = Plan(_lib.MFCC if mfcc else _lib.MEL, device, window_length=len(w), step_length=h, layout=layout,
                 n_filters=fb.shape[0], n_coefs=ncoef, row_align=row_align, f64=f64, with_mel=also_mel)
        p.set_window(w)
        p.set_mel_filterbank(fb)
        if mfcc:
            p.set_dct(constants.dct2_rows(n_filters, ncoef))
        return p
    return _cached(key, make)


def mel_mfcc_supported(window_length, number_filters, number_coefficients, f64=False):
    """Geometries of the one-pass melspectrogram + mfcc kernel (k_mel2 MODE 4)."""
    return (not f64) and window_length == 2048 and 1 <= number_filters <= 128 and 1 <= number_coefficients <= min(32, number_filters - 1)


def _cqt_f32_max_bins(fft_length):
    """Rows of a kernel matrix the float32 k_cqt holds at this fft_length (zafx_cqt_max_bins); 0 outside its sizes."""
    n = ctypes.c_int(0)
    _lib.check(_lib.load().zafx_cqt_max_bins(int(fft_length), ctypes.byref(n)), "zafx_cqt_max_bins")
    return n.value


_CQT_MIN_FFT = 512   # shortest frame of the device kernels


def _cqt_embed(cqt_kernel, step, target=_CQT_MIN_FFT):
    """A kernel whose fft_length L is below the device kernels' minimum, rewritten for frames of `target` samples.

    K . fft(frame) = G . frame with G = fft(K, axis=1) (the kernel's rows in the time domain, zaf.py:630-632), so the same
    numbers come out of K' . fft(frame') for the longer frame' when G' holds G at the offset d by which the longer frame
    starts earlier (left padding ceil((L - step) / 2) -> ceil((target - step) / 2), zaf.py:612-620) and zeros elsewhere,
    and K' = ifft(G', axis=1).  K' is dense (target columns per row): small, since L < target."""
    import scipy.sparse
    k = np.asarray(cqt_kernel.toarray(), dtype=np.complex128)
    length = k.shape[1]
    d = -((step - target) // 2) - -((step - length) // 2)   # ceil((target - step) / 2) - ceil((L - step) / 2)
    g = np.zeros((k.shape[0], target), np.complex128)
    g[:, d:d + length] = np.fft.fft(k, axis=1)
    return scipy.sparse.csr_matrix(np.fft.ifft(g, axis=1))


def cqt_plan(sampling_frequency, time_resolution, cqt_kernel, octave_resolution=None, layout="FT", device=0, row_align=0, f64=False):
    if not hasattr(cqt_kernel, "tocsr"):
        raise ValueError("cqt_kernel must be a scipy.sparse matrix (as returned by cqtkernel)")
    n_bins, fft_length = cqt_kernel.shape
    step = round(sampling_frequency / time_resolution)   # zaf.py:603 (banker's rounding)
    if step < 1:
        raise ValueError("time_resolution too high for this sampling_frequency")
    if step > fft_length:   # (the reference's right padding floor((fft_length - step) / 2) is negative there: np.pad raises, zaf.py:612-620)
        raise ValueError("step (sampling_frequency / time_resolution) exceeds the kernel's fft_length")
    if 2 <= fft_length < _CQT_MIN_FFT:
        # (a short kernel -- few bins per octave, high minimum frequency -- of any length, power of two or not)
        cqt_kernel = _cqt_embed(cqt_kernel, step)
        fft_length = _CQT_MIN_FFT
    if fft_length < _CQT_MIN_FFT or fft_length > 131072 or fft_length & (fft_length - 1):
        raise ValueError(f"zafx CQT kernels need a power-of-two fft_length in [512, 131072] (or any length below 512), got {fft_length}")
    # a frame above 65536 samples runs on the float64 kernel, which decimates the frame; so do kernels with more rows than
    # fit beside the frame (k_cqt keeps the whole frame + one output column of the kernel's rows in the 160 KB of LDS:
    # about 4 000 rows at fft_length 32768 -- the reference's own 208-row example is far inside)
    csr = cqt_kernel.tocsr()
    # fft_length 65536 (minimum frequencies down to 23 Hz at 44.1 kHz, 24 bins per octave) runs on the float32 kernel as the even and
    # the odd bins of two 16384-point transforms when the matrix touches the low bins 1 ... 8191 (and their mirrors) only
    low_band = True
    if fft_length == 65536 and csr.nnz:
        m = np.minimum(csr.indices, fft_length - csr.indices)
        low_band = bool(m.min() >= 1 and m.max() <= 8191)
    f64 = bool(f64) or fft_length > 65536 or not low_band or n_bins > _cqt_f32_max_bins(fft_length)
    chroma = octave_resolution is not None
    key = ("chroma" if chroma else "cqt", device, fft_length, step, n_bins, int(octave_resolution or 0), _LAYOUTS[layout],
           _as_row_align(row_align, layout), bool(f64),
           _digest(csr.indptr, csr.indices, csr.data))

    def make():
        p = Plan(_lib.CHROMA if chroma else _lib.CQT, device, step_length=step, layout=layout, fft_length=fft_length,
                 n_bins=n_bins, octave_resolution=int(octave_resolution or 0), row_align=row_align, f64=f64)
        p.set_cqt_kernel(csr)
        return p
    return _cached(key, make)


def linear_plan(matrix, device=0):
    """y = matrix @ x for every clip (the carrier of the dct / dst transforms)."""
    m = np.ascontiguousarray(matrix, dtype=np.float64)
    if m.ndim != 2 or not (1 <= m.shape[0] <= 16384 and 1 <= m.shape[1] <= 16384):
        raise ValueError("matrix must be 2-D with both sides in [1, 16384]")
    key = ("linear", device, m.shape, _digest(m))

    def make():
        p = Plan(_lib.LINEAR, device, window_length=m.shape[1], n_filters=m.shape[0])
        p.set_matrix(m)
        return p
    return _cached(key, make)


# ======================================================================================
# batched API (build-defined extension): (clips, samples) float32 in, float32/complex64 out
# ======================================================================================
def _run_host_into(plan, x, n_in, out):
    """Plan.run_host with the caller's `out` honoured even when the library promoted the plan to float64 (windows under 33
    samples, more than 576 filters, a CQT outside the float32 kernel's reach): the float64 result is then cast into `out`
    -- which is returned -- instead of raising on the dtype."""
    if out is None or np.dtype(out.dtype) == plan.out_dtype:
        return plan.run_host(x, n_in, out=out)
    res = plan.run_host(x, n_in)
    if tuple(out.shape) != tuple(res.shape) or not out.flags.writeable:
        raise ValueError(f"out must be a writeable array of shape {tuple(res.shape)}")
    np.copyto(out, res, casting="same_kind")
    return out


_ROW_PADDING = {"value": "auto"}


def set_row_padding(mode):
    """How the *_batch functions of the STFT / MDCT families lay out the device (F, T) array when the caller passes neither `out` nor
    `row_align`: "auto" (default) pads the rows to whole 128-byte lines whenever T is off that grid and hands back a view of the padded result
    (_line_grid); "compact" always keeps the reference's own memory order (C-contiguous results, the kernels' off-grid forms)."""
    if mode not in ("auto", "compact"):
        raise ValueError('row padding must be "auto" or "compact"')
    _ROW_PADDING["value"] = mode


def get_row_padding():
    return _ROW_PADDING["value"]


def _line_grid(plan, n_in, out, padded_plan, frames=None, row_align=None):
    """The plan a *_batch call of the STFT / MDCT families runs: when the frame count T is off the 128-byte line grid of the (F, T) rows in the
    reference layout (the usual case: T % 16 != 0 for complex64 rows), the device array gets padded rows (row_align = one line) -- the kernels then
    run at the rate they have on the grid (T = 433 against 432: ISTFT 0.47 -> 0.63 of HBM, |X| rows 0.34 -> 0.49, MDCT 0.46 -> 0.58)
    -- and the NumPy array handed back is a VIEW of the padded result: the reference's shape, dtype and indexing, rows `pitch` elements apart
    instead of T (C-contiguity of the result is not part of the reference's contract; np.ascontiguousarray(result) gives the compact
    array).  A caller's own `out` array, the frame-major layout and frame counts on the grid keep the compact plan.  row_align (of the *_batch
    functions): None = this rule, 0 = always the compact device array (the reference's own memory order), n = rows padded to n elements."""
    if row_align is not None:
        return padded_plan(int(row_align)) if int(row_align) > 1 and plan.layout == _lib.LAYOUT_FT else plan
    if out is not None or plan.layout != _lib.LAYOUT_FT or plan.row_align > 1 or _ROW_PADDING["value"] != "auto":
        return plan
    t = plan.out_dims(n_in)[1] if frames is None else int(frames)
    elem = plan.out_dtype if frames is None else plan.in_dtype   # the dtype of the 2-D side
    a = _line_elements(elem)
    return padded_plan(a) if t % a else plan


def stft_batch(clips, window_function, step_length, layout="FT", device=0, onesided=False, f64=False, out=None, row_align=None):
    """(B, N) -> (B, W, T) complex64 [layout "FT"] or (B, T, W) ["TF"].

    out (every *_batch function): destination array of the result's shape and dtype, e.g. a reused zafx.pinned_empty
    array -- with page-locked arrays on both sides the chunked, double-buffered transfer (Plan.run_host) runs at the PCIe rate.
    When the library computes a float32 request in float64 (see Plan.f64) the result is cast into `out` afterwards.

    onesided=True keeps rows 0..W/2 only -- what every example of the reference slices out of the
    result (zaf.py:83) -- and halves the bytes written; onesided="magnitude" / "power" returns |X| / |X|^2
    of those rows as a real array."""
    x = _as_clips(clips, dtype=np.float64 if f64 else np.float32)
    plan = _line_grid(stft_plan(window_function, step_length, layout, device, onesided, f64), x.shape[1], out,
                      lambda a: stft_plan(window_function, step_length, layout, device, onesided, f64, row_align=a), row_align=row_align)
    out = _run_host_into(plan, x.astype(plan.in_dtype, copy=False), x.shape[1], out)
    if f64 or not plan.f64:
        return out
    return out.astype(np.complex64 if np.iscomplexobj(out) else np.float32, copy=False)   # (computed in float64: window not a power of two)


def istft_batch(spectra, window_function, step_length, layout="FT", device=0, onesided=False, f64=False, out=None, row_align=None):
    """(B, W, T) ["FT"] or (B, T, W) ["TF"] complex -> (B, T*H - (W-H)) float32.

    onesided=True takes rows 0..W/2 and completes X[W-k] = conj X[k]: the result equals the two-sided
    call on the spectrum of a real signal."""
    w = _as_window(window_function, any_length=True)
    s = np.asarray(spectra)
    if s.ndim != 3:
        raise ValueError("spectra must be 3-D")
    wl, nt = (s.shape[1], s.shape[2]) if _LAYOUTS[layout] == _lib.LAYOUT_FT else (s.shape[2], s.shape[1])
    if wl != (len(w) // 2 + 1 if onesided else len(w)):
        raise ValueError("spectrum rows must equal window_length (window_length/2 + 1 when onesided)")
    plan = _line_grid(istft_plan(w, step_length, layout, device, onesided, f64), nt, None,
                      lambda a: istft_plan(w, step_length, layout, device, onesided, f64, row_align=a), frames=nt, row_align=row_align)
    out = _run_host_into(plan, s if plan.row_align > 1 else np.ascontiguousarray(s, dtype=plan.in_dtype), nt, out)
    return out if f64 else out.astype(np.float32, copy=False)   # (a very small hop is computed in float64 whatever f64 says)


def mdct_batch(clips, window_function, layout="FT", device=0, f64=False, out=None, row_align=None):
    """(B, N) -> (B, W/2, T) float32 ["FT"] or (B, T, W/2) ["TF"]; f64: float64 arrays and arithmetic."""
    x = _as_clips(clips, dtype=np.float64 if f64 else np.float32)
    plan = _line_grid(mdct_plan(window_function, layout, device, f64=f64), x.shape[1], out,
                      lambda a: mdct_plan(window_function, layout, device, row_align=a, f64=f64), row_align=row_align)
    out = _run_host_into(plan, x.astype(plan.in_dtype, copy=False), x.shape[1], out)
    return out if f64 else out.astype(np.float32, copy=False)


def imdct_batch(coefficients, window_function, layout="FT", device=0, f64=False, out=None, row_align=None):
    """(B, W/2, T) ["FT"] or (B, T, W/2) ["TF"] -> (B, (W/2)(T-1) - 1) float32 (float64 with f64)."""
    c = np.asarray(coefficients)
    if c.dtype != (np.float64 if f64 else np.float32):
        c = c.astype(np.float64 if f64 else np.float32)
    w = _as_window(window_function, any_length=True)
    if c.ndim != 3:
        raise ValueError("coefficients must be 3-D")
    nf, nt = (c.shape[1], c.shape[2]) if _LAYOUTS[layout] == _lib.LAYOUT_FT else (c.shape[2], c.shape[1])
    if 2 * nf != len(w):
        raise ValueError("coefficient rows must equal window_length/2")
    plan = _line_grid(mdct_plan(w, layout, device, inverse=True, f64=f64), nt, None,
                      lambda a: mdct_plan(w, layout, device, inverse=True, row_align=a, f64=f64), frames=nt, row_align=row_align)
    out = _run_host_into(plan, c if plan.row_align > 1 else c.astype(plan.in_dtype, copy=False), nt, out)
    return out if f64 else out.astype(np.float32, copy=False)


def melspectrogram_batch(clips, window_function, step_length, mel_filterbank, layout="FT", device=0, f64=False, out=None):
    """(B, N) -> (B, n_filters, T) float32 (float64 arrays and arithmetic with f64)."""
    x = _as_clips(clips, dtype=np.float64 if f64 else np.float32)   # (validated before any device call)
    plan = mel_plan(window_function, step_length, mel_filterbank, None, layout, device, f64=f64)
    x = x.astype(plan.in_dtype, copy=False)
    out = _run_host_into(plan, x, x.shape[1], out)
    return out if f64 else out.astype(np.float32, copy=False)   # (a long window is computed in float64 whatever f64 says)


def mfcc_batch(clips, window_function, step_length, mel_filterbank, number_coefficients, layout="FT", device=0, f64=False, out=None):
    """(B, N) -> (B, number_coefficients, T) float32 (float64 arrays and arithmetic with f64)."""
    x = _as_clips(clips, dtype=np.float64 if f64 else np.float32)   # (validated before any device call)
    plan = mel_plan(window_function, step_length, mel_filterbank, number_coefficients, layout, device, f64=f64)
    x = x.astype(plan.in_dtype, copy=False)
    out = _run_host_into(plan, x, x.shape[1], out)
    return out if f64 else out.astype(np.float32, copy=False)


def _split_mel_mfcc(both, n_filters, layout):
    """(mel, mfcc) views of the one-pass plan's output: rows 0 .. n_filters - 1 and the rest (the last axis in the frame-major layout)."""
    if _LAYOUTS[layout] == _lib.LAYOUT_FT:
        return both[:, :n_filters], both[:, n_filters:]
    return both[:, :, :n_filters], both[:, :, n_filters:]


def mel_mfcc_batch(clips, window_function, step_length, mel_filterbank, number_coefficients, layout="FT", device=0, f64=False, out=None):
    """zaf.melspectrogram AND zaf.mfcc of the same clips from ONE set of transforms (both start with the same zaf.stft, zaf.py:369 and :436;
    BASELINE config 3): (B, N) -> ((B, n_filters, T), (B, number_coefficients, T)), views of one (B, n_filters + number_coefficients, T)
    array (`out`, when given, is that array); both bit-identical to melspectrogram_batch / mfcc_batch.  Geometries outside the one-pass
    kernel (mel_mfcc_supported: float32, window 2048, <= 128 filters, <= 32 coefficients) run the two plans."""
    x = _as_clips(clips, dtype=np.float64 if f64 else np.float32)
    w = _as_window(window_function, any_length=True)
    n_filters = mel_filterbank.shape[0] if hasattr(mel_filterbank, "shape") else 0
    if not mel_mfcc_supported(len(w), n_filters, int(number_coefficients), f64):
        mel = melspectrogram_batch(x, w, step_length, mel_filterbank, layout, device, f64)
        cep = mfcc_batch(x, w, step_length, mel_filterbank, number_coefficients, layout, device, f64)
        if out is None:
            return mel, cep
        np.copyto(out, np.concatenate([mel, cep], axis=1 if _LAYOUTS[layout] == _lib.LAYOUT_FT else 2), casting="same_kind")
        return _split_mel_mfcc(out, n_filters, layout)
    plan = mel_plan(w, step_length, mel_filterbank, number_coefficients, layout, device, also_mel=True)
    both = _run_host_into(plan, x, x.shape[1], out)
    return _split_mel_mfcc(both, n_filters, layout)


def mel_mfcc_pcm_batch(pcm, window_function, step_length, mel_filterbank, number_coefficients, layout="FT", device=0, out=None):
    """mel_mfcc_batch of integer PCM clips (see stft_pcm_batch): int16 read inside the kernel's own loads."""
    w = _as_window(window_function, any_length=True)
    n_filters = mel_filterbank.shape[0]
    if not mel_mfcc_supported(len(w), n_filters, int(number_coefficients)):
        return mel_mfcc_batch(pcm_to_mono(pcm, device), w, step_length, mel_filterbank, number_coefficients, layout, device, out=out)
    plan = mel_plan(w, step_length, mel_filterbank, number_coefficients, layout, device, also_mel=True)
    return _split_mel_mfcc(plan.run_host_pcm(pcm, out=out), n_filters, layout)


def cqtspectrogram_batch(clips, sampling_frequency, time_resolution, cqt_kernel, layout="FT", device=0, f64=False, out=None):
    """(B, N) -> (B, n_bins, T) float32 (float64 arrays and arithmetic with f64)."""
    x = _as_clips(clips, dtype=np.float64 if f64 else np.float32)   # (validated before any device call)
    plan = cqt_plan(sampling_frequency, time_resolution, cqt_kernel, None, layout, device, f64=f64)
    x = x.astype(plan.in_dtype, copy=False)
    out = _run_host_into(plan, x, x.shape[1], out)
    return out if f64 else out.astype(np.float32, copy=False)   # (a long kernel is computed in float64 whatever f64 says)


def cqtchromagram_batch(clips, sampling_frequency, time_resolution, octave_resolution, cqt_kernel, layout="FT", device=0, f64=False, out=None):
    """(B, N) -> (B, octave_resolution, T) float32 (float64 arrays and arithmetic with f64)."""
    x = _as_clips(clips, dtype=np.float64 if f64 else np.float32)   # (validated before any device call)
    plan = cqt_plan(sampling_frequency, time_resolution, cqt_kernel, int(octave_resolution), layout, device, f64=f64)
    x = x.astype(plan.in_dtype, copy=False)
    out = _run_host_into(plan, x, x.shape[1], out)
    return out if f64 else out.astype(np.float32, copy=False)


def _pcm_device_mono(plan, pcm):
    """Upload integer PCM (clips, frames[, channels]) and return the normalised mono f32 DeviceBuffer."""
    pcm = np.ascontiguousarray(pcm)
    if pcm.dtype not in (np.dtype(np.int16), np.dtype(np.int32)):
        raise ValueError("PCM ingest takes int16 or int32 samples (wavread's other dtypes: convert on the host)")
    if pcm.ndim == 2:
        pcm = pcm[:, :, None]
    if pcm.ndim != 3:
        raise ValueError("pcm must be (clips, frames) or (clips, frames, channels)")
    b, n, ch = pcm.shape
    d_pcm = DeviceBuffer.from_host(pcm, plan.device)
    d_x = DeviceBuffer((b, n), np.float32, plan.device)
    plan.pcm_to_float(d_pcm, d_x, b, n, ch)
    return d_pcm, d_x


def _pcm_batch(plan, batch_fn, pcm, out):
    """A *_pcm_batch call: the chunked three-stream pipeline with integer uploads (Plan.run_host_pcm); plans the library runs in
    float64 (windows outside the float32 kernels) normalise on the device and take the general path, `out` included."""
    if plan.f64:
        return batch_fn(pcm_to_mono(pcm, plan.device), out)
    return plan.run_host_pcm(pcm, out=out)


def stft_pcm_batch(pcm, window_function, step_length, layout="FT", device=0, onesided=False, out=None):
    """STFT of integer PCM clips (clips, frames[, channels]): wavread's x / 2^(bits-1) and the channel mean
    (zaf.py:1202, :65) run on the device in front of the transform; only 2-4 B per sample and channel cross PCIe."""
    plan = stft_plan(window_function, step_length, layout, device, onesided)
    return _pcm_batch(plan, lambda x, o: stft_batch(x, window_function, step_length, layout, device, onesided, out=o), pcm, out)


def mdct_pcm_batch(pcm, window_function, layout="FT", device=0, out=None):
    """mdct_batch of integer PCM clips (see stft_pcm_batch)."""
    plan = mdct_plan(window_function, layout, device)
    return _pcm_batch(plan, lambda x, o: mdct_batch(x, window_function, layout, device, out=o), pcm, out)


def melspectrogram_pcm_batch(pcm, window_function, step_length, mel_filterbank, layout="FT", device=0, out=None):
    """melspectrogram_batch of integer PCM clips (see stft_pcm_batch): 2 bytes per sample up, 4 n_filters / hop down."""
    plan = mel_plan(window_function, step_length, mel_filterbank, None, layout, device)
    return _pcm_batch(plan, lambda x, o: melspectrogram_batch(x, window_function, step_length, mel_filterbank, layout, device, out=o), pcm, out)


def mfcc_pcm_batch(pcm, window_function, step_length, mel_filterbank, number_coefficients, layout="FT", device=0, out=None):
    """mfcc_batch of integer PCM clips (see stft_pcm_batch)."""
    plan = mel_plan(window_function, step_length, mel_filterbank, number_coefficients, layout, device)
    return _pcm_batch(plan, lambda x, o: mfcc_batch(x, window_function, step_length, mel_filterbank, number_coefficients, layout, device, out=o), pcm, out)


def cqtspectrogram_pcm_batch(pcm, sampling_frequency, time_resolution, cqt_kernel, layout="FT", device=0, out=None):
    """cqtspectrogram_batch of integer PCM clips (see stft_pcm_batch)."""
    plan = cqt_plan(sampling_frequency, time_resolution, cqt_kernel, None, layout, device)
    return _pcm_batch(plan, lambda x, o: cqtspectrogram_batch(x, sampling_frequency, time_resolution, cqt_kernel, layout, device, out=o), pcm, out)


def cqtchromagram_pcm_batch(pcm, sampling_frequency, time_resolution, octave_resolution, cqt_kernel, layout="FT", device=0, out=None):
    """cqtchromagram_batch of integer PCM clips (see stft_pcm_batch)."""
    plan = cqt_plan(sampling_frequency, time_resolution, cqt_kernel, int(octave_resolution), layout, device)
    return _pcm_batch(plan, lambda x, o: cqtchromagram_batch(x, sampling_frequency, time_resolution, octave_resolution, cqt_kernel, layout, device, out=o), pcm, out)


def pcm_to_mono(pcm, device=0):
    """(clips, frames[, channels]) int16/int32 -> (clips, frames) float32: mean over channels of x / 2^(bits-1)."""
    plan = stft_plan(constants.hamming(64), 32, "FT", device)   # any plan provides the stream
    with plan.lock:
        d_pcm, d_x = _pcm_device_mono(plan, pcm)
        try:
            plan.sync()
            return d_x.download()
        finally:
            d_pcm.free()
            d_x.free()


def dct_fft_length(length, kind, sine):
    """M, the complex FFT length zafx_dct.hip runs for a dct / dst of this length and type -- or None when the length is
    outside that kernel (M must be a power of two in [32, 8192]): N/2 for types 2-4, N-1 (dct) / N+1 (dst) for type 1."""
    n = int(length)
    m = (n + 1 if sine else n - 1) if kind == 1 else (n // 2 if n % 2 == 0 else 0)
    return m if 32 <= m <= 8192 and m & (m - 1) == 0 else None


DCT_CHIRP_MAX = 8192   # longest vector of the chirp-z form (its convolution of 2^ceil(log2(2N - 1)) <= 16384 points lives in LDS)


def dct_plan(length, kind, sine=False, device=0):
    """Plan of the orthonormal dct (sine=False) / dst of type `kind` of vectors of `length` samples (zaf.py:703-839, :842-981).
    Lengths with N/2 (type 1: N -/+ 1) a power of two in [32, 8192] run ONE M-point complex transform per vector (k_dct, instead of the
    reference's 2N-2 ... 8N-point one); every other length from 2 to 8192 runs on the Bluestein machinery -- types 2-4 of a length 4 j as the
    same maps around an N/2-point convolution (k_dct_bsh: two transforms of 2^ceil(log2(N - 1)) points), the rest as a chirp-z sum over all N
    points (k_dct_bs32: two of 2^ceil(log2(2N - 1))) -- O(N log N) for every length the reference takes."""
    n = int(length)
    if dct_fft_length(n, kind, sine) is None and not 2 <= n <= DCT_CHIRP_MAX:
        raise ValueError("dct / dst plans take lengths 2 ... 8192, and longer ones whose N/2 (type 1: N-1 / N+1) is a power of two up to 8192")
    return _cached(("dct_fft", n, int(kind), bool(sine), device),
                   lambda: Plan(_lib.DCT, device, window_length=n, transform_type=int(kind), transform_sine=bool(sine)))


def _transform_batch(vectors, matrix_fn, kind, device, out=None):
    x = np.ascontiguousarray(vectors, dtype=np.float32)
    if x.ndim != 2 or x.shape[1] < 1:
        raise ValueError("vectors must be 2-D (batch, length) with length >= 1")
    n = x.shape[1]
    if kind not in (1, 2, 3, 4):
        raise ValueError("type must be 1, 2, 3 or 4")
    sine = matrix_fn is constants.dst_matrix
    if dct_fft_length(n, kind, sine) is not None or 2 <= n <= DCT_CHIRP_MAX:   # O(N log N): on the FFT core, as the reference computes it
        return dct_plan(n, kind, sine, device).run_host(x, n, out=out)
    # what is left -- one sample, or 8192 < N <= 16384 off the power-of-two grid (a convolution of 32768 points does not fit LDS) --: the
    # transform as a dense matrix on the matrix cores, O(N^2) arithmetic and an N x N table, so bounded
    if n > 16384:
        raise ValueError("dct / dst lengths above 16384 are not supported (above 8192: N/2, or N-1 / N+1 for type 1, must be a power of two)")

    def make():   # the N x N matrix (O(N^2) trigonometry, 8 N^2 bytes) is built once per (transform, type, N, device)
        m = np.ascontiguousarray(matrix_fn(n, kind), dtype=np.float64)
        p = Plan(_lib.LINEAR, device, window_length=m.shape[1], n_filters=m.shape[0])
        p.set_matrix(m)
        return p
    plan = _cached((matrix_fn.__name__, int(kind), n, device), make)
    return plan.run_host(x, n, out=out)


def dct_batch(vectors, dct_type, device=0, out=None):
    """(B, N) -> (B, N) float32: orthonormal DCT of type 1-4 of every row (zaf.dct per row)."""
    return _transform_batch(vectors, constants.dct_matrix, dct_type, device, out)


def dst_batch(vectors, dst_type, device=0, out=None):
    """(B, N) -> (B, N) float32: orthonormal DST of type 1-4 of every row (zaf.dst per row)."""
    return _transform_batch(vectors, constants.dst_matrix, dst_type, device, out)


def dct(audio_signal, dct_type):
    """Drop-in for zaf.dct (zaf.py:703): (N,) -> (N,) float64, type 1, 2, 3 or 4."""
    x = np.asarray(audio_signal)
    if x.ndim != 1:
        raise ValueError("audio_signal must be 1-D; use dct_batch for (batch, length)")
    return dct_batch(x[None, :], dct_type)[0].astype(np.float64)


def dst(audio_signal, dst_type):
    """Drop-in for zaf.dst (zaf.py:842): (N,) -> (N,) float64, type 1, 2, 3 or 4."""
    x = np.asarray(audio_signal)
    if x.ndim != 1:
        raise ValueError("audio_signal must be 1-D; use dst_batch for (batch, length)")
    return dst_batch(x[None, :], dst_type)[0].astype(np.float64)


# ======================================================================================
# drop-in API: the zaf.* signatures (1 clip, float64 / complex128 results)
# ======================================================================================
def stft(audio_signal, window_function, step_length):
    """Drop-in for zaf.stft (zaf.py:45): (N,) -> (W, T) complex128, two-sided."""
    f64 = _PRECISION["value"] == "f64"
    x = _as_signal(audio_signal, np.float64 if f64 else np.float32)
    return stft_batch(x, window_function, step_length, f64=f64)[0].astype(np.complex128)


def istft(audio_stft, window_function, step_length):
    """Drop-in for zaf.istft (zaf.py:144): (W, T) -> (T*H - (W-H),) float64."""
    s = np.asarray(audio_stft)
    if s.ndim != 2:
        raise ValueError("audio_stft must be 2-D (window_length, number_times)")
    return istft_batch(s[None], window_function, step_length, f64=_PRECISION["value"] == "f64")[0].astype(np.float64)


def melspectrogram(audio_signal, window_function, step_length, mel_filterbank):
    """Drop-in for zaf.melspectrogram (zaf.py:324): (N,) -> (n_filters, T) float64."""
    f64 = _PRECISION["value"] == "f64"
    x = _as_signal(audio_signal, np.float64 if f64 else np.float32)
    return melspectrogram_batch(x, window_function, step_length, mel_filterbank, f64=f64)[0].astype(np.float64)


def mfcc(audio_signal, window_function, step_length, mel_filterbank, number_coefficients):
    """Drop-in for zaf.mfcc (zaf.py:378): (N,) -> (number_coefficients, T) float64."""
    f64 = _PRECISION["value"] == "f64"
    x = _as_signal(audio_signal, np.float64 if f64 else np.float32)
    return mfcc_batch(x, window_function, step_length, mel_filterbank, number_coefficients, f64=f64)[0].astype(np.float64)


def cqtspectrogram(audio_signal, sampling_frequency, time_resolution, cqt_kernel):
    """Drop-in for zaf.cqtspectrogram (zaf.py:562): (N,) -> (n_bins, T) float64."""
    f64 = _PRECISION["value"] == "f64"
    x = _as_signal(audio_signal, np.float64 if f64 else np.float32)
    return cqtspectrogram_batch(x, sampling_frequency, time_resolution, cqt_kernel, f64=f64)[0].astype(np.float64)


def cqtchromagram(audio_signal, sampling_frequency, time_resolution, octave_resolution, cqt_kernel):
    """Drop-in for zaf.cqtchromagram (zaf.py:638): (N,) -> (octave_resolution, T) float64."""
    f64 = _PRECISION["value"] == "f64"
    x = _as_signal(audio_signal, np.float64 if f64 else np.float32)
    return cqtchromagram_batch(x, sampling_frequency, time_resolution, octave_resolution, cqt_kernel, f64=f64)[0].astype(np.float64)


def mdct(audio_signal, window_function):
    """Drop-in for zaf.mdct (zaf.py:984): (N,) -> (W/2, T) float64."""
    f64 = _PRECISION["value"] == "f64"
    x = _as_signal(audio_signal, np.float64 if f64 else np.float32)
    return mdct_batch(x, window_function, f64=f64)[0].astype(np.float64)


def imdct(audio_mdct, window_function):
    """Drop-in for zaf.imdct (zaf.py:1078): (W/2, T) -> ((W/2)(T-1) - 1,) float64."""
    c = np.asarray(audio_mdct)
    if c.ndim != 2:
        raise ValueError("audio_mdct must be 2-D (number_frequencies, number_times)")
    return imdct_batch(c[None], window_function, f64=_PRECISION["value"] == "f64")[0].astype(np.float64)
