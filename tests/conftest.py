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
        for name in ("tiny", "consts", "config", "lengths", "dctdst", "cqtfull")
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
