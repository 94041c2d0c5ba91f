"""Host-side mirror of the reference interface (seqoia.h:336-374) over the C-ABI.

``encode`` / ``decode`` / ``write`` / ``read`` take and return the same things
as sqoa_encode / sqoa_decode / sqoa_write / sqoa_read: a failure the reference
signals with NULL / 0 is ``None`` / ``0`` here.  ``EncodePlan`` / ``DecodePlan``
wrap the batch entry points of sqoa_hip.h for device-resident torch tensors.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import DECODE, ENCODE, Desc, Job, Result, SqoaError, check

SQOA_SRGB, SQOA_LINEAR = 0, 1
ST_OK, ST_BADARG, ST_OVERFLOW, ST_INTERNAL, ST_SLOWPATH = 0, 1, 2, 3, 4


def stride_channels(channels: int) -> int:
    """Input bytes per pixel for desc.channels (reference seqoia.h:476-486): 5/6 read as 3/4."""
    return (1 if channels < 3 else 3) + (0 if channels & 1 else 1)


# ---- legacy single-image API ---------------------------------------------------------------


def encode(pixels, desc: Desc) -> Optional[bytes]:
    """sqoa_encode (seqoia.h:363): host pixels -> complete stream, or None."""
    lib = _lib.load()
    buf = np.ascontiguousarray(pixels, dtype=np.uint8)
    n = C.c_int(0)
    p = lib.sqoa_encode(buf.ctypes.data if buf.size else None, C.byref(desc), C.byref(n))
    if not p:
        return None
    out = C.string_at(p, n.value)
    _lib.c_free(p)
    return out


def decode(data, channels: int = 0) -> Tuple[Optional[np.ndarray], Desc]:
    """sqoa_decode (seqoia.h:374): stream -> (flat uint8 pixels or None, header desc)."""
    lib = _lib.load()
    raw = bytes(data)
    d = Desc()
    p = lib.sqoa_decode(raw, len(raw), C.byref(d), channels)
    if not p:
        return None, d
    oc = channels if channels else stride_channels(d.channels)
    out = np.frombuffer(C.string_at(p, d.width * d.height * oc), dtype=np.uint8).copy()
    _lib.c_free(p)
    return out, d


def write(filename: str, pixels, desc: Desc) -> int:
    """sqoa_write (seqoia.h:336): returns bytes written, 0 on failure."""
    buf = np.ascontiguousarray(pixels, dtype=np.uint8)
    return _lib.load().sqoa_write(filename.encode(), buf.ctypes.data, C.byref(desc))


def read(filename: str, channels: int = 0) -> Tuple[Optional[np.ndarray], Desc]:
    """sqoa_read (seqoia.h:350)."""
    lib = _lib.load()
    d = Desc()
    p = lib.sqoa_read(filename.encode(), C.byref(d), channels)
    if not p:
        return None, d
    oc = channels if channels else stride_channels(d.channels)
    out = np.frombuffer(C.string_at(p, d.width * d.height * oc), dtype=np.uint8).copy()
    _lib.c_free(p)
    return out, d


# ---- batch API on device-resident buffers ---------------------------------------------------


def encode_bound(desc: Desc) -> int:
    return _lib.load().sqoa_hip_encode_bound(C.byref(desc))


class Plan:
    """sqoa_hip_plan: a job table resident in HBM, launched any number of times."""

    def __init__(self, kind: int, jobs: Sequence[Job], device: int = 0):
        self.lib = _lib.load()
        self.n = len(jobs)
        self.kind = kind
        self.device = device
        self._jobs = (Job * max(self.n, 1))(*jobs)
        self._h = C.c_void_p()
        check(self.lib.sqoa_hip_plan_create(device, kind, self._jobs, self.n, C.byref(self._h)), "sqoa_hip_plan_create")

    def update(self, jobs: Sequence[Job]):
        """sqoa_hip_plan_update: a new job table (at most as many jobs as the plan was created for) without re-allocating."""
        jobs = list(jobs)
        arr = (Job * max(len(jobs), 1))(*jobs)
        check(self.lib.sqoa_hip_plan_update(self._h, arr, len(jobs)), "sqoa_hip_plan_update")
        self._jobs, self.n = arr, len(jobs)

    def launch(self, stream: int = 0):
        check(self.lib.sqoa_hip_plan_launch(self._h, C.c_void_p(stream) if stream else None), "sqoa_hip_plan_launch")

    def results(self) -> np.ndarray:
        """Waits for the plan's stream; structured array with fields len, status, width, height, channels, colorspace, qoi_compat."""
        res = (Result * max(self.n, 1))()
        check(self.lib.sqoa_hip_plan_results(self._h, res, self.n), "sqoa_hip_plan_results")
        dt = np.dtype([("len", "<i4"), ("status", "<i4"), ("width", "<u4"), ("height", "<u4"), ("channels", "u1"),
                       ("colorspace", "u1"), ("qoi_compat", "u1"), ("pad", "u1")])
        return np.frombuffer(bytes(res), dtype=dt, count=self.n).copy()

    def shape(self) -> int:
        """sqoa_hip_plan_shape: the kernel shape the next launch takes (decode plans; bit 3 = the balanced kernel)."""
        return self.lib.sqoa_hip_plan_shape(self._h)

    def time(self, iters: int) -> float:
        """Milliseconds for `iters` back-to-back launches, HIP events on the plan's own stream."""
        ms = C.c_float(0)
        check(self.lib.sqoa_hip_plan_time(self._h, iters, C.byref(ms)), "sqoa_hip_plan_time")
        return ms.value

    def pack(self, stream: int = 0):
        """Encode plans: tightly packed blob of all streams + int64 offsets [n+1] (torch CUDA tensors)."""
        import torch

        dev = torch.device("cuda", self.device)
        res = self.results()
        cap = int(res["len"].astype(np.int64).sum())
        blob = torch.empty((max(cap, 1),), dtype=torch.uint8, device=dev)
        offsets = torch.empty((self.n + 1,), dtype=torch.int64, device=dev)
        total = C.c_ulonglong(0)
        check(self.lib.sqoa_hip_pack(self._h, blob.data_ptr(), cap, offsets.data_ptr(), C.byref(total),
                                     C.c_void_p(stream) if stream else None), "sqoa_hip_pack")
        return blob[: total.value], offsets

    def close(self):
        if self._h:
            self.lib.sqoa_hip_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def _device_index(t) -> int:
    return t.device.index if t.device.index is not None else 0


def _batch_pitch(t) -> int:
    """Image pitch of a [n, h, w, c] batch whose images are each contiguous (the batch itself may be padded)."""
    n, h, w, c = t.shape
    assert t.stride()[1:] == (w * c, c, 1) and (n <= 1 or t.stride(0) >= h * w * c), "images must be contiguous"
    return t.stride(0) if n > 1 else h * w * c


class EncodePlan(Plan):
    """Encode a batch of same-shaped images held in one torch uint8 CUDA tensor [n, h, w, c]
    (contiguous, or any view whose images are each contiguous: stride(0) is the image pitch).

    Output: one worst-case slot per image in ``self.out`` ([n, slot] uint8 on the
    same device); ``results()['len']`` gives the stream length of each.
    """

    def __init__(self, pixels, qoi_compat: int = 0, colorspace: int = 0, channels: Optional[int] = None):
        import torch

        assert pixels.is_cuda and pixels.dtype == torch.uint8 and pixels.dim() == 4
        n, h, w, c = pixels.shape
        pitch = _batch_pitch(pixels)
        desc = Desc(w, h, channels if channels is not None else c, colorspace, qoi_compat)
        assert stride_channels(desc.channels) == c
        self.pixels = pixels
        self.slot = encode_bound(desc)
        self.out = torch.empty((n, self.slot), dtype=torch.uint8, device=pixels.device)
        src0, dst0 = pixels.data_ptr(), self.out.data_ptr()
        jobs = [Job(src0 + i * pitch, dst0 + i * self.slot, 0, self.slot, desc, 0) for i in range(n)]
        super().__init__(ENCODE, jobs, _device_index(pixels))

    def streams(self, idx=None):
        """D2H the encoded streams (all, or the given indices) as bytes objects."""
        res = self.results()
        idx = range(self.n) if idx is None else idx
        return [bytes(self.out[i, : int(res["len"][i])].cpu().numpy()) for i in idx]


class EncodeList(Plan):
    """Encode images of different shapes in one plan (BASELINE config 5 shape: a suite of mixed sizes).

    ``images``: list of (torch uint8 CUDA tensor holding h*w*stride bytes, Desc).  All output
    slots live in one buffer ``self.out`` at ``self.offsets``; ``streams()`` copies them out.
    """

    def __init__(self, images):
        import torch

        dev = images[0][0].device
        self.images = images
        self.slots = [encode_bound(d) for _, d in images]
        self.offsets = [0]
        for sl in self.slots:
            self.offsets.append(self.offsets[-1] + sl)
        self.out = torch.empty((max(self.offsets[-1], 1),), dtype=torch.uint8, device=dev)
        base = self.out.data_ptr()
        jobs = []
        for (t, d), off, sl in zip(images, self.offsets, self.slots):
            assert t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()
            jobs.append(Job(t.data_ptr(), base + off, 0, min(sl, 0xFFFFFFFF), d, 0))
        super().__init__(ENCODE, jobs, _device_index(images[0][0]))

    def streams(self):
        res = self.results()
        return [None if res["status"][i] else bytes(self.out[o : o + int(res["len"][i])].cpu().numpy())
                for i, o in enumerate(self.offsets[:-1])]


class DecodePlan(Plan):
    """Decode n streams held in device memory.

    ``streams``: torch uint8 CUDA tensor [n, slot] (e.g. EncodePlan.out) with
    ``lens`` bytes used in each row; every image decodes to width*height*out_channels
    bytes in ``self.out`` [n, h, w, out_channels] (``out``: a caller-provided batch, possibly with a larger image pitch).
    """

    def __init__(self, streams, lens, width: int, height: int, out_channels: int, channels_arg: Optional[int] = None, out=None):
        import torch

        assert streams.is_cuda and streams.dtype == torch.uint8 and streams.is_contiguous() and streams.dim() == 2
        n, slot = streams.shape
        self.streams = streams
        if out is None:
            out = torch.empty((n, height, width, out_channels), dtype=torch.uint8, device=streams.device)
        assert out.is_cuda and out.dtype == torch.uint8 and tuple(out.shape) == (n, height, width, out_channels)
        self.out = out
        pitch = _batch_pitch(out)
        arg = out_channels if channels_arg is None else channels_arg
        desc = Desc(width, height, arg, 0, 0)
        src0, dst0, img = streams.data_ptr(), self.out.data_ptr(), height * width * out_channels
        jobs = [Job(src0 + i * slot, dst0 + i * pitch, int(lens[i]), img, desc, 0) for i in range(n)]
        super().__init__(DECODE, jobs, _device_index(streams))


class DecodeList(Plan):
    """Decode streams of different shapes in one plan (BASELINE config 5 shape).

    ``items``: list of (device address of the stream, stream bytes, width, height, out_channels).  Every image decodes
    to width*height*out_channels bytes at ``self.offsets[i]`` of ``self.out`` (offsets are multiples of 4, so that
    4-channel images are stored as dwords)."""

    def __init__(self, items, device: int = 0, channels_arg=None):
        import torch

        self.items = list(items)
        self.offsets = [0]
        for _, _, w, h, oc in self.items:
            self.offsets.append(self.offsets[-1] + ((w * h * oc + 3) & ~3))
        self.out = torch.empty((max(self.offsets[-1], 1),), dtype=torch.uint8, device=torch.device("cuda", device))
        base = self.out.data_ptr()
        jobs = []
        for (src, size, w, h, oc), off in zip(self.items, self.offsets):
            arg = oc if channels_arg is None else channels_arg
            jobs.append(Job(src, base + off, int(size), min(w * h * oc, 0xFFFFFFFF), Desc(w, h, arg, 0, 0), 0))
        super().__init__(DECODE, jobs, device)

    def image(self, i):
        """Decoded pixels of image i as a flat uint8 CUDA tensor (a view of self.out)."""
        _, _, w, h, oc = self.items[i]
        return self.out[self.offsets[i] : self.offsets[i] + w * h * oc]


def fnv1a64(ptrs, lens, device: int = 0):
    """FNV-1a-64 of device buffers (one digest per buffer), computed on the device: size-independent parity checks."""
    n = len(ptrs)
    p = (C.c_ulonglong * max(n, 1))(*[int(x) for x in ptrs])
    l = (C.c_uint * max(n, 1))(*[int(x) for x in lens])
    out = (C.c_ulonglong * max(n, 1))()
    check(_lib.load().sqoa_hip_fnv1a64(device, p, l, n, out), "sqoa_hip_fnv1a64")
    return [int(out[i]) for i in range(n)]


def synth(out, cls: str = "mixed", seed: int = 1, img0: int = 0):
    """Fill a torch uint8 CUDA tensor [n, h, w, c] (c in 3,4) with the synthetic images of SURVEY.md Appendix C."""
    n, h, w, c = out.shape
    assert out.is_cuda
    kinds = {"mixed": 0, "icon": 1, "photo": 2}
    check(_lib.load().sqoa_hip_synth(_device_index(out), out.data_ptr(), _batch_pitch(out), w, h, c, kinds[cls], seed, img0, n), "sqoa_hip_synth")
    return out
