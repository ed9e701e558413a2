"""ctypes binding of libzafx.so (the C-ABI declared in include/zafx.h).

There is NO CPU fallback: if the shared library is missing this module raises at
import of the first symbol, and every transform call fails loudly.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))

# enum zafx_kind
STFT, ISTFT, MDCT, IMDCT, MEL, MFCC, CQT, CHROMA, LINEAR, DCT = 1, 2, 3, 4, 5, 6, 7, 8, 9, 10
# enum zafx_layout
LAYOUT_FT, LAYOUT_TF = 0, 1
# enum zafx_spectrum
SPECTRUM_TWO_SIDED, SPECTRUM_ONE_SIDED, SPECTRUM_MAGNITUDE, SPECTRUM_POWER = 0, 1, 2, 3
# enum zafx_precision
PRECISION_F32, PRECISION_F64 = 0, 1
# enum zafx_constant
CONST_WINDOW, CONST_MEL_FB, CONST_DCT, CONST_CQT_INDPTR, CONST_CQT_INDICES, CONST_CQT_VALUES, CONST_MATRIX = 1, 2, 3, 4, 5, 6, 7


# return code of zafx_alloc when the device is out of memory (ZAFX_ERROR_OUT_OF_MEMORY in include/zafx.h = hipErrorOutOfMemory)
ERROR_OUT_OF_MEMORY = 2


class ZafxParams(ctypes.Structure):
    _fields_ = [
        ("struct_size", ctypes.c_int32),
        ("window_length", ctypes.c_int32),
        ("step_length", ctypes.c_int32),
        ("layout", ctypes.c_int32),
        ("n_filters", ctypes.c_int32),
        ("n_coefs", ctypes.c_int32),
        ("fft_length", ctypes.c_int32),
        ("n_bins", ctypes.c_int32),
        ("octave_resolution", ctypes.c_int32),
        ("spectrum", ctypes.c_int32),
        ("precision", ctypes.c_int32),
        ("row_align", ctypes.c_int32),
        ("transform_type", ctypes.c_int32),
        ("transform_sine", ctypes.c_int32),
        ("with_mel", ctypes.c_int32),
        ("reserved", ctypes.c_int32 * 1),
    ]


class ZafxError(RuntimeError):
    """A libzafx call returned a non-zero code."""


# every symbol include/zafx.h declares: name -> (restype, argtypes)
_vp, _i, _i64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t
SYMBOLS = {
    "zafx_version": (_i, []),
    "zafx_last_error": (ctypes.c_char_p, []),
    "zafx_device_count": (_i, [ctypes.POINTER(_i)]),
    "zafx_device_name": (_i, [_i, ctypes.c_char_p, _sz]),
    "zafx_alloc": (_i, [_i, ctypes.POINTER(_vp), _sz]),
    "zafx_free": (_i, [_i, _vp]),
    "zafx_alloc_placed": (_i, [_vp, ctypes.POINTER(_vp), _sz, _vp, _i64, _i64, _i, _i, ctypes.POINTER(ctypes.c_float)]),
    "zafx_memset": (_i, [_i, _vp, _i, _sz]),
    "zafx_h2d": (_i, [_i, _vp, _vp, _sz]),
    "zafx_d2h": (_i, [_i, _vp, _vp, _sz]),
    "zafx_d2d": (_i, [_i, _vp, _vp, _sz]),
    "zafx_host_alloc": (_i, [ctypes.POINTER(_vp), _sz]),
    "zafx_host_free": (_i, [_vp]),
    "zafx_plan_create": (_i, [ctypes.POINTER(_vp), _i, _i, ctypes.POINTER(ZafxParams)]),
    "zafx_plan_destroy": (_i, [_vp]),
    "zafx_plan_set_constant": (_i, [_vp, _i, _vp, _sz]),
    "zafx_plan_out_dims": (_i, [_vp, _i64, ctypes.POINTER(_i64)]),
    "zafx_plan_row_pitch": (_i, [_vp, _i64, ctypes.POINTER(_i64)]),
    "zafx_execute": (_i, [_vp, _vp, _vp, _i64, _i64]),
    "zafx_sync": (_i, [_vp]),
    "zafx_plan_clip_bytes": (_i, [_vp, _i64, ctypes.POINTER(_i64), ctypes.POINTER(_i64)]),
    "zafx_run_host": (_i, [_vp, _vp, _vp, _i64, _i64, _i64]),
    "zafx_run_host_pcm": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _i, _i64]),
    "zafx_timer_start": (_i, [_vp]),
    "zafx_timer_stop": (_i, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "zafx_plan_kernel_name": (_i, [_vp, ctypes.c_char_p, _sz]),
    "zafx_plan_last_kernel_name": (_i, [_vp, ctypes.c_char_p, _sz]),
    "zafx_cqt_max_bins": (_i, [_i, ctypes.POINTER(_i)]),
    "zafx_pcm_to_float": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _i]),
    "zafx_execute_pcm": (_i, [_vp, _vp, _vp, _i64, _i64, _i, _i]),
    "zafx_comm_unique_id": (_i, [_vp]),
    "zafx_comm_create": (_i, [ctypes.POINTER(_vp), _i, _i, _i, _vp]),
    "zafx_comm_destroy": (_i, [_vp]),
    "zafx_comm_count": (_i, [_vp, ctypes.POINTER(_i)]),
    "zafx_comm_user_rank": (_i, [_vp, ctypes.POINTER(_i)]),
    "zafx_comm_broadcast_constants": (_i, [_vp, _vp, _i]),
}

_lib = None


def library_path():
    return os.environ.get("ZAFX_LIBRARY", os.path.join(_HERE, "libzafx.so"))


def load():
    """Load libzafx.so once; raise ZafxError (never fall back) if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = library_path()
    if not os.path.exists(path):
        raise ZafxError(
            f"libzafx.so not found at {path}: build it with `python __graft_entry__.py build` "
            "(or `make -C zaf-python_amd/csrc`). zafx has no CPU fallback."
        )
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.zafx_version() != 101:
        raise ZafxError(f"libzafx.so version {lib.zafx_version()} does not match the Python binding (101)")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().zafx_last_error()
        raise ZafxError(f"{what}: {msg.decode() if msg else 'error'} (code {rc})")


def device_count():
    n = ctypes.c_int(0)
    check(load().zafx_device_count(ctypes.byref(n)), "zafx_device_count")
    return n.value


def device_name(device=0):
    buf = ctypes.create_string_buffer(256)
    check(load().zafx_device_name(device, buf, 256), "zafx_device_name")
    return buf.value.decode()
