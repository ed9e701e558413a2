"""INTEGRATION.md section 2 -- the ctypes stub a maintainer would paste into zaf.py -- executed VERBATIM.

CPU part: the stub's parameter struct is the library's (field for field, byte for byte).  GPU part (-m gpu): the fenced Python block is
extracted from the document and exec'd -- only `ctypes.CDLL("libzafx.so")` is pointed at the in-tree library -- and its `stft` is held to the
reference's own outputs (tests/golden/tiny.npz, config.npz probes) at the float32 tolerance.
"""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT, relerr, synth_clip


def stub_source():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    section = text[text.index("## 2. Patching `zaf.py` itself"):text.index("## 3. Build and deployment")]
    blocks = re.findall(r"```python\n(.*?)```", section, flags=re.S)
    assert len(blocks) == 1, "section 2 holds exactly one Python block: the stub"
    return blocks[0]


def load_stub(monkeypatch, library):
    real = ctypes.CDLL
    monkeypatch.setattr(ctypes, "CDLL", lambda name, *a, **k: real(library if name == "libzafx.so" else name, *a, **k))
    ns = {}
    exec(compile(stub_source(), "INTEGRATION.md#2", "exec"), ns)
    return ns


def test_stub_struct_is_the_library_struct(monkeypatch, built_library):
    from zafx import _lib
    ns = load_stub(monkeypatch, built_library)
    stub, ours = ns["_Params"], _lib.ZafxParams
    assert ctypes.sizeof(stub) == ctypes.sizeof(ours)
    assert [(n, getattr(stub, n).offset, getattr(stub, n).size) for n, _ in stub._fields_] == \
           [(n, getattr(ours, n).offset, getattr(ours, n).size) for n, _ in ours._fields_]
    assert (ns["ZAFX_STFT"], ns["ZAFX_LAYOUT_FT"], ns["ZAFX_CONST_WINDOW"]) == (_lib.STFT, _lib.LAYOUT_FT, _lib.CONST_WINDOW)
    header = open(os.path.join(ROOT, "include", "zafx.h")).read()
    body = header[header.index("typedef struct zafx_params {"):header.index("} zafx_params;")]
    declared = re.findall(r"^\s*int32_t\s+(\w+)", body, flags=re.M)
    assert declared == [n for n, _ in stub._fields_]   # include/zafx.h declares the same fields in the same order


@pytest.mark.gpu
def test_stub_stft_against_the_reference(monkeypatch, built_library, golden):
    ns = load_stub(monkeypatch, built_library)
    stft = ns["stft"]
    g = golden["tiny"]
    for n in (1, 63, 64, 65, 1000):
        for hop in (32, 16):
            got = stft(g[f"x_{n}"], g["ham"], hop)
            assert got.dtype == np.complex128 and got.shape == g[f"stft_{n}_{hop}"].shape
            assert relerr(got, g[f"stft_{n}_{hop}"]) <= 1e-5
    # BASELINE config 1 geometry: one 10 s clip of white noise, Hamming 2048 / 1024, against the probes of the real reference's output
    cfg = golden["config"]
    x = synth_clip(0, 0, 441000)
    ham = 0.54 - 0.46 * np.cos(2 * np.pi * np.arange(2048) / 2048)   # scipy.signal.windows.hamming(2048, sym=False)
    got = stft(x, ham, 1024)
    assert tuple(cfg["S0_stft_shape"]) == got.shape == (2048, 432)
    assert np.max(np.abs(got.reshape(-1)[cfg["S0_stft_idx"]] - cfg["S0_stft_val"])) <= 1e-5 * float(cfg["S0_stft_maxabs"])
