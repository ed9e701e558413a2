"""frame_sync (zafx_fft.hpp) exchanges a frame between the lanes of ONE wavefront through LDS without s_waitcnt:
the hardware serves a wave's DS instructions in issue order.  This test guards that assumption against toolchain
changes: the library built with -DZAFX_WAVE_SYNC_FENCE (`make -C zaf-python_amd/csrc fence`, also done by
__graft_entry__.build) must produce bit-identical outputs for every transform."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PKG = os.path.join(ROOT, "zaf-python_amd", "zafx")


@pytest.mark.timeout(900)
def test_fenced_build_is_bit_identical(tmp_path):
    fence = os.path.join(PKG, "libzafx_fence.so")
    if not os.path.exists(fence):
        subprocess.run(["make", "-C", os.path.join(ROOT, "zaf-python_amd", "csrc"), "fence", "-j", "8"], check=True, stdout=subprocess.DEVNULL)
    outs = {}
    for tag, lib in (("default", os.path.join(PKG, "libzafx.so")), ("fence", fence)):
        path = str(tmp_path / f"{tag}.npz")
        env = dict(os.environ, ZAFX_LIBRARY=lib)
        res = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fence_probe.py"), path], env=env, capture_output=True, timeout=400)
        assert res.returncode == 0, res.stderr.decode()[-2000:]
        assert os.path.basename(lib) in res.stdout.decode()
        outs[tag] = np.load(path)
    assert sorted(outs["default"].files) == sorted(outs["fence"].files) and len(outs["default"].files) > 80
    for key in outs["default"].files:
        a, b = outs["default"][key], outs["fence"][key]
        assert a.shape == b.shape and a.dtype == b.dtype and np.array_equal(a.view(np.uint8), b.view(np.uint8)), key
        assert np.all(np.isfinite(a)), key
