#!/usr/bin/env python3
"""Encode time of the headline batch as a function of the byte distance between consecutive images / output slots,
inside ONE source arena and ONE slot arena (same physical memory for every setting).  Workgroups advance through
their images in step, so power-of-two image strides put all of them on the same address bits at the same time."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seqoia_amd as sq
from seqoia_amd.api import Plan, ENCODE, DECODE, Job
ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=1024); ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--pads", default="0,0x100,0x1000,0x10000,0x100000,0x11100,0x111100"); ap.add_argument("--spads", default="0,0x11100")
ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
n, s = a.images, a.size
img = s * s * 4
desc = sq.Desc(s, s, 4, 0, 0)
bound = sq.encode_bound(desc)
pads = [int(x, 0) for x in a.pads.split(",")]; spads = [int(x, 0) for x in a.spads.split(",")]
px = torch.empty((n, s, s, 4), dtype=torch.uint8, device="cuda")
sq.synth(px, "mixed", 1, 0)
src = torch.empty(n * (img + max(pads)), dtype=torch.uint8, device="cuda")
dst = torch.empty(n * (bound + max(spads)), dtype=torch.uint8, device="cuda")
print(f"src arena {src.data_ptr():#x}, slot arena {dst.data_ptr():#x}")
class Raw(Plan):
    def __init__(self, kind, jobs): super().__init__(kind, jobs, 0)
for pad in pads:
    st = img + pad
    src[: n * st].view(n, st)[:, :img].copy_(px.view(n, img))
    for spad in spads:
        sl = bound + spad
        enc = Raw(ENCODE, [Job(src.data_ptr() + i * st, dst.data_ptr() + i * sl, 0, sl, desc, 0) for i in range(n)])
        enc.launch(); res = enc.results(); assert (res["status"] == 0).all()
        ms = enc.time(a.iters) / a.iters
        print(f"image stride {st:#9x} slot stride {sl:#9x}: encode {ms:.3f} ms/launch", flush=True)
        enc.close()
# decode: streams from the slots of a plain encode, pixel output at the same image strides
enc = sq.EncodePlan(px); enc.launch(); lens = enc.results()["len"]
for pad in pads:
    st = img + pad
    dec = Raw(DECODE, [Job(enc.out.data_ptr() + i * enc.slot, src.data_ptr() + i * st, int(lens[i]), img, desc, 0) for i in range(n)])
    dec.launch(); res = dec.results(); assert (res["status"] == 0).all()
    ms = dec.time(a.iters) / a.iters
    print(f"output stride {st:#9x}: decode {ms:.3f} ms/launch", flush=True)
    dec.close()
