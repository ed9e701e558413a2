"""The C-ABI shared library loads and exports every symbol include/zafx.h declares
(no compute calls: this runs on machines without a GPU)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT

HEADER = os.path.join(ROOT, "include", "zafx.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(zafx_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    from zafx import _lib
    assert declared_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(built_library):
    lib = ctypes.CDLL(built_library)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.zafx_version.restype = ctypes.c_int
    assert lib.zafx_version() == 101


def test_params_struct_layout(built_library):
    from zafx import _lib
    assert ctypes.sizeof(_lib.ZafxParams) == 16 * 4   # 12 fields + 4 reserved int32


def test_errors_are_reported_not_swallowed(built_library):
    """Without a GPU every entry point that needs one returns a code and a message."""
    from zafx import _lib
    lib = _lib.load()
    n = ctypes.c_int(-1)
    rc = lib.zafx_device_count(ctypes.byref(n))
    if rc == 0 and n.value > 0:
        pytest.skip("a GPU is present")
    assert rc != 0 or n.value == 0
    if rc != 0:
        assert lib.zafx_last_error()
    bad = _lib.ZafxParams()
    h = ctypes.c_void_p()
    assert lib.zafx_plan_create(ctypes.byref(h), 0, _lib.STFT, ctypes.byref(bad)) != 0   # struct_size mismatch
    assert b"struct_size" in lib.zafx_last_error()


def test_missing_library_fails_loudly():
    """No CPU fallback: a missing libzafx.so is an error, never a silent NumPy path."""
    code = (
        "import os, sys; os.environ['ZAFX_LIBRARY'] = '/nonexistent/libzafx.so';"
        f"sys.path.insert(0, {os.path.join(ROOT, 'zaf-python_amd')!r});"
        "import numpy as np, zafx;\n"
        "try:\n"
        "    zafx.stft(np.zeros(4096, np.float32), zafx.hamming(2048), 1024)\n"
        "except zafx.ZafxError as e:\n"
        "    print('LOUD', e)\n"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert "LOUD" in out.stdout and "no CPU fallback" in out.stdout, out.stdout + out.stderr


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "zaf-python_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".cpp", ".hpp", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"(from|import)\s+oracle|zaf_oracle|oracle/", text), os.path.join(dirpath, f)
