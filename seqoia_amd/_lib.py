"""ctypes binding of libsqoa_mi355x.so -- the C-ABI boundary (include/seqoia.h, include/sqoa_hip.h).

The library is the product; this module only declares its symbols.  There is no
Python or CPU implementation behind it: if the shared object is missing the
import fails, and if no gfx950 device is present every compute call fails.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
# SQOA_LIB: developer override to A/B kernel variants; the default is the in-tree build
LIB_PATH = os.environ.get("SQOA_LIB") or os.path.join(HERE, "lib", "libsqoa_mi355x.so")


class Desc(C.Structure):
    """sqoa_desc (reference seqoia.h:318-324), sizeof == 12."""

    _fields_ = [
        ("width", C.c_uint),
        ("height", C.c_uint),
        ("channels", C.c_ubyte),
        ("colorspace", C.c_ubyte),
        ("qoi_compat", C.c_ubyte),
    ]

    def __repr__(self):
        return f"Desc({self.width}x{self.height}, channels={self.channels}, colorspace={self.colorspace}, qoi_compat={self.qoi_compat})"


class Job(C.Structure):
    """sqoa_hip_job, 40 bytes."""

    _fields_ = [
        ("src", C.c_ulonglong),
        ("dst", C.c_ulonglong),
        ("src_size", C.c_uint),
        ("dst_cap", C.c_uint),
        ("desc", Desc),
        ("reserved", C.c_uint),
    ]


class Result(C.Structure):
    """sqoa_hip_result, 20 bytes."""

    _fields_ = [("len", C.c_int), ("status", C.c_int), ("desc", Desc)]


assert C.sizeof(Desc) == 12 and C.sizeof(Job) == 40 and C.sizeof(Result) == 20

ENCODE, DECODE = 0, 1

# every symbol include/seqoia.h and include/sqoa_hip.h declare: (restype, argtypes)
SYMBOLS = {
    "sqoa_encode": (C.c_void_p, [C.c_void_p, C.POINTER(Desc), C.POINTER(C.c_int)]),
    "sqoa_decode": (C.c_void_p, [C.c_void_p, C.c_int, C.POINTER(Desc), C.c_int]),
    "sqoa_write": (C.c_int, [C.c_char_p, C.c_void_p, C.POINTER(Desc)]),
    "sqoa_read": (C.c_void_p, [C.c_char_p, C.POINTER(Desc), C.c_int]),
    "sqoa_hip_set_allocator": (None, [C.c_void_p, C.c_void_p]),
    "sqoa_hip_device_count": (C.c_int, []),
    "sqoa_hip_lds_order": (C.c_int, [C.c_int]),
    "sqoa_hip_lds_order_probe": (C.c_int, [C.c_int, C.POINTER(C.c_uint)]),
    "sqoa_hip_last_error": (C.c_char_p, []),
    "sqoa_hip_encode_bound": (C.c_size_t, [C.POINTER(Desc)]),
    "sqoa_hip_malloc": (C.c_int, [C.c_int, C.c_size_t, C.POINTER(C.c_ulonglong)]),
    "sqoa_hip_free": (C.c_int, [C.c_int, C.c_ulonglong]),
    "sqoa_hip_memcpy_h2d": (C.c_int, [C.c_int, C.c_ulonglong, C.c_void_p, C.c_size_t]),
    "sqoa_hip_memcpy_d2h": (C.c_int, [C.c_int, C.c_void_p, C.c_ulonglong, C.c_size_t]),
    "sqoa_hip_host_alloc": (C.c_int, [C.c_size_t, C.POINTER(C.c_void_p)]),
    "sqoa_hip_host_free": (C.c_int, [C.c_void_p]),
    "sqoa_hip_stream_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "sqoa_hip_stream_destroy": (C.c_int, [C.c_int, C.c_void_p]),
    "sqoa_hip_stream_sync": (C.c_int, [C.c_int, C.c_void_p]),
    "sqoa_hip_memcpy_h2d_async": (C.c_int, [C.c_int, C.c_ulonglong, C.c_void_p, C.c_size_t, C.c_void_p]),
    "sqoa_hip_memcpy_d2h_async": (C.c_int, [C.c_int, C.c_void_p, C.c_ulonglong, C.c_size_t, C.c_void_p]),
    "sqoa_hip_plan_stream": (C.c_void_p, [C.c_void_p]),
    "sqoa_hip_plan_shape": (C.c_int, [C.c_void_p]),
    "sqoa_hip_encode_route": (C.c_int, [C.POINTER(C.c_ulonglong), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sqoa_hip_decode_route": (C.c_int, [C.POINTER(C.c_uint), C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "sqoa_hip_plan_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(Job), C.c_int, C.POINTER(C.c_void_p)]),
    "sqoa_hip_plan_destroy": (None, [C.c_void_p]),
    "sqoa_hip_plan_update": (C.c_int, [C.c_void_p, C.POINTER(Job), C.c_int]),
    "sqoa_hip_plan_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "sqoa_hip_plan_results": (C.c_int, [C.c_void_p, C.POINTER(Result), C.c_int]),
    "sqoa_hip_plan_time": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "sqoa_hip_encode_batch": (C.c_int, [C.c_int, C.POINTER(Job), C.c_int, C.POINTER(Result)]),
    "sqoa_hip_decode_batch": (C.c_int, [C.c_int, C.POINTER(Job), C.c_int, C.POINTER(Result)]),
    "sqoa_hip_pack": (C.c_int, [C.c_void_p, C.c_ulonglong, C.c_size_t, C.c_ulonglong, C.POINTER(C.c_ulonglong), C.c_void_p]),
    "sqoa_hip_synth": (C.c_int, [C.c_int, C.c_ulonglong, C.c_size_t, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_uint, C.c_uint, C.c_int]),
    "sqoa_hip_fnv1a64": (C.c_int, [C.c_int, C.POINTER(C.c_ulonglong), C.POINTER(C.c_uint), C.c_int, C.POINTER(C.c_ulonglong)]),
}

_lib = None


def load() -> C.CDLL:
    """Load the shared library; raises if it has not been built (run `make lib`)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make lib` (or __graft_entry__.build()); there is no fallback path")
        # A process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64 / libhsa-runtime64; if this library
        # were loaded first it would pull in the system copies, and whichever copy initialises the device second finds none
        # (sqoa_hip_device_count() == 0).  So where PyTorch exists, its runtime is loaded first and this library binds to it.
        try:
            import torch  # noqa: F401
        except Exception:  # noqa: BLE001 -- no PyTorch: the system ROCm runtime is the only one
            pass
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]
_libc.free.restype = None


def c_free(ptr):
    _libc.free(ptr)


class SqoaError(RuntimeError):
    pass


def check(rc: int, what: str):
    if rc != 0:
        raise SqoaError(f"{what} failed ({rc}): {load().sqoa_hip_last_error().decode()}")
