"""pytest configuration: markers, import paths, shared input generators."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "zaf-python_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


def synth_clip(seed, c, n):
    """SURVEY 8(d) synthetic input: white Gaussian noise, sigma 1, f32-rounded."""
    return np.random.default_rng([seed, c]).standard_normal(n).astype(np.float32)


@pytest.fixture(scope="session")
def golden():
    return {
        name: np.load(os.path.join(GOLDEN, name + ".npz"))
        for name in ("tiny", "consts", "config", "lengths", "dctdst", "cqtfull", "signals")
    }


def relerr(out, ref):
    """Normwise error of SURVEY 8(d): max|out - ref| / max|ref| (complex modulus)."""
    out = np.asarray(out)
    ref = np.asarray(ref)
    assert out.shape == ref.shape, (out.shape, ref.shape)
    denom = np.max(np.abs(ref)) if ref.size else 1.0
    if denom == 0:
        denom = 1.0
    return float(np.max(np.abs(out - ref)) / denom) if ref.size else 0.0


@pytest.fixture(scope="session")
def built_library():
    """Path of libzafx.so; builds it with hipcc (cross-compile, no GPU needed) if absent."""
    import subprocess
    lib = os.path.join(ROOT, "zaf-python_amd", "zafx", "libzafx.so")
    if not os.path.exists(lib):
        subprocess.run(["make", "-C", os.path.join(ROOT, "zaf-python_amd", "csrc"), "-j", "8"], check=True,
                       stdout=subprocess.DEVNULL)
    return lib


# ---------------------------------------------------------------- row-wise bounds (tests/signals.py cases)
def excess(out, ref, bound):
    """max |out - ref| / bound, elementwise; 0 where both are 0 (an exact result), inf where only the bound is."""
    d = np.abs(np.asarray(out) - np.asarray(ref))
    b = np.broadcast_to(np.asarray(bound, dtype=np.float64), d.shape)
    r = np.zeros(d.shape)
    nz = d > 0
    with np.errstate(divide="ignore"):
        r[nz] = d[nz] / b[nz]
    return float(r.max()) if r.size else 0.0


def row_bound(ref, tol, floor):
    """The per-row contract: |out - ref| <= 10 tol max_row|ref| + floor.  `floor` (scalar, per row or per element) is the
    level below which float32 arithmetic cannot follow the float64 reference: a row above it is held to 10 tol of its
    OWN level, whatever the loudest row of the clip is."""
    ref = np.asarray(ref)
    rows = np.abs(ref).max(axis=-1, keepdims=True) if ref.ndim > 1 else np.abs(ref)
    return 10.0 * tol * rows + floor


def bin_noise(spec_ref, c, eps):
    """Error model of one spectrum: every bin of frame t carries at most c eps max_k|X[k, t]| (a transform's rounding
    errors scale with the operands of its butterflies, and for a tonal frame some bins are the difference of two
    partial sums as large as the peak).  (1, T)."""
    return c * eps * np.abs(spec_ref).max(axis=0, keepdims=True)


def dct2_rows(m, lo, hi):
    """Rows lo..hi-1 of the orthonormal DCT-II of length m (scipy.fftpack.dct(norm='ortho'), zaf.py:443-449)."""
    k = np.arange(lo, hi)[:, None]
    n = np.arange(m)[None, :]
    d = np.sqrt(2.0 / m) * np.cos(np.pi * (2 * n + 1) * k / (2 * m))
    d[k[:, 0] == 0] /= np.sqrt(2.0)
    return d


def mfcc_floor(spec_ref, fb_dense, ncoef, c, eps, c_bin=3.0):
    """What float arithmetic of epsilon `eps` can do to zaf.mfcc (zaf.py:436-452) on this very signal.  (ncoef, T).

    Two parts.  (i) The transform's rounding errors land on every bin of a frame at about the same absolute level, c_bin eps sqrt(log2 W)
    times the RMS of the frame's (two-sided) spectrum -- for a tone that is its peak / sqrt(W / 2), for noise about its level --, independent
    from bin to bin: the band powers move by the root-sum-square of FB (2 |X| nu + nu^2), the logs by the width of that interval (unbounded
    relative to a band the reference itself only holds as round-off), the coefficients by the root-sum-square of D width.  (ii) The
    pipeline's own roundings behind the transform -- band sums, logarithms, the DCT's dot products over the log levels --, c eps each,
    summed linearly.  Round 6: (i) used to be c eps of the frame's PEAK on every bin, summed linearly through filterbank and DCT -- a bound
    1e3 times above the measured errors on tones off the bin grid; now the worst measured coefficient of a signal sits at 0.13 (tones off
    the grid) ... 0.65 (tone on a bin) of it: `error_over_floor` in the report of tests/test_gpu_signals.py."""
    w = 2 * (spec_ref.shape[0] - 1)
    mag = np.abs(spec_ref[1:fb_dense.shape[1] + 1])
    two_sided = np.concatenate([np.abs(spec_ref) ** 2, np.abs(spec_ref[-2:0:-1]) ** 2], axis=0)
    nu = c_bin * eps * np.sqrt(np.log2(w)) * np.sqrt(two_sided.mean(axis=0, keepdims=True))
    band = fb_dense @ (mag ** 2)
    dband = np.sqrt((fb_dense ** 2) @ ((2.0 * mag * nu + nu ** 2) ** 2)) + c * eps * band
    e = np.finfo(float).eps
    logref = np.log(band + e)
    width = np.maximum(np.log(band + dband + e) - logref, logref - np.log(np.maximum(band - dband, 0.0) + e))
    d = np.abs(dct2_rows(fb_dense.shape[0], 1, ncoef + 1))
    return 1.5 * np.sqrt((d ** 2) @ (width ** 2)) + c * eps * (d @ np.abs(logref))   # (1.5: the widths of neighbouring floor-level bands are not independent)
