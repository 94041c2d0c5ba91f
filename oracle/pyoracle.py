"""ctypes loaders for the checker libraries -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  Two checkers are exposed with one interface:

* ``restatement()`` -- oracle/libsqoa_oracle.so, the CPU restatement
  (oracle/sqoa_oracle.c).
* ``reference()``   -- oracle/_ref/libsqoa_ref.so, the real reference compiled
  by oracle/Makefile from /root/reference/seqoia.h (``ref_sqoa_*`` symbols,
  seqoia.h:336-374).  ``None`` when it has not been built.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))

CLASSES = {"mixed": 0, "icon": 1, "photo": 2}


class Desc(C.Structure):
    """Field-for-field sqoa_desc (seqoia.h:318-324); sizeof == 12."""

    _fields_ = [
        ("width", C.c_uint),
        ("height", C.c_uint),
        ("channels", C.c_ubyte),
        ("colorspace", C.c_ubyte),
        ("qoi_compat", C.c_ubyte),
    ]


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]
_libc.free.restype = None


def build() -> None:
    """Compile the restatement and (when /root/reference exists) oracle/_ref."""
    subprocess.run(["make", "-s", "-C", HERE], check=True)


class Codec:
    """encode/decode through a library exporting <prefix>encode / <prefix>decode."""

    def __init__(self, lib: C.CDLL, prefix: str, kind: str):
        self.kind = kind
        self.lib = lib
        self._enc = getattr(lib, prefix + "encode")
        self._dec = getattr(lib, prefix + "decode")
        self._enc.argtypes = [C.c_void_p, C.POINTER(Desc), C.POINTER(C.c_int)]
        self._enc.restype = C.c_void_p
        self._dec.argtypes = [C.c_void_p, C.c_int, C.POINTER(Desc), C.c_int]
        self._dec.restype = C.c_void_p

    def encode(self, pixels, width, height, channels, qoi_compat=0, colorspace=0):
        """-> bytes, or None where the codec returns NULL."""
        buf = np.ascontiguousarray(pixels, dtype=np.uint8)
        d = Desc(width, height, channels, colorspace, qoi_compat)
        n = C.c_int(0)
        p = self._enc(buf.ctypes.data, C.byref(d), C.byref(n))
        if not p:
            return None
        out = C.string_at(p, n.value)
        _libc.free(p)
        return out

    def encode_raw(self, ptr, desc, n):
        """Timing helper: encode + free with caller-made ctypes arguments."""
        p = self._enc(ptr, desc, n)
        _libc.free(p)

    def decode(self, data, channels=0):
        """-> (pixels uint8 ndarray [h*w*out_channels], Desc) or (None, Desc)."""
        raw = bytes(data)
        d = Desc()
        p = self._dec(raw, len(raw), C.byref(d), channels)
        if not p:
            return None, d
        col = 1 if d.channels < 3 else 3
        oc = channels if channels else col + (1 if d.channels % 2 == 0 else 0)
        n = d.width * d.height * oc
        out = np.frombuffer(C.string_at(p, n), dtype=np.uint8).copy()
        _libc.free(p)
        return out, d

    def decode_raw(self, ptr, size, desc, channels):
        p = self._dec(ptr, size, desc, channels)
        _libc.free(p)


_cache = {}


def restatement() -> Codec:
    if "orc" not in _cache:
        path = os.path.join(HERE, "libsqoa_oracle.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        lib.orc_synth.argtypes = [C.c_void_p, C.c_uint, C.c_uint, C.c_uint, C.c_int, C.c_uint32, C.c_uint32]
        lib.orc_synth.restype = None
        lib.orc_fnv1a64.argtypes = [C.c_void_p, C.c_size_t]
        lib.orc_fnv1a64.restype = C.c_uint64
        lib.orc_encode_bound.argtypes = [C.c_uint, C.c_uint, C.c_uint]
        lib.orc_encode_bound.restype = C.c_size_t
        _cache["orc"] = Codec(lib, "orc_", "port")
    return _cache["orc"]


def reference():
    if "ref" not in _cache:
        path = os.path.join(HERE, "_ref", "libsqoa_ref.so")
        if not os.path.exists(path) and os.path.exists("/root/reference/seqoia.h"):
            build()
        _cache["ref"] = Codec(C.CDLL(path), "ref_sqoa_", "reference") if os.path.exists(path) else None
    return _cache["ref"]


def best() -> Codec:
    """The real reference when built, else the restatement."""
    return reference() or restatement()


def synth(width, height, channels, cls="mixed", seed=1, img=0) -> np.ndarray:
    """Synthetic image of SURVEY.md Appendix C as a flat uint8 array."""
    out = np.empty(width * height * channels, dtype=np.uint8)
    restatement().lib.orc_synth(out.ctypes.data, width, height, channels, CLASSES[cls], seed, img)
    return out


def fnv1a64(data) -> int:
    raw = bytes(data)
    return restatement().lib.orc_fnv1a64(raw, len(raw))
