#!/usr/bin/env python3
"""Launch only the product kernels on the headline batch (for rocprofv3 --pmc passes).
    python tools/prof_kernels.py [--images 256] [--iters 3] [--what encode|decode|both] [--qoi 0|1]"""
import argparse, os, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seqoia_amd as sq

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=256)
ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--iters", type=int, default=3)
ap.add_argument("--what", default="both")
ap.add_argument("--qoi", type=int, default=0)
ap.add_argument("--cls", default="mixed")
ap.add_argument("--channels", type=int, default=4)
ap.add_argument("--out-channels", type=int, default=4)
a = ap.parse_args()
px = torch.empty((a.images, a.size, a.size, a.channels), dtype=torch.uint8, device="cuda")
sq.synth(px, a.cls, 1, 0)
enc = sq.EncodePlan(px, qoi_compat=a.qoi)
enc.launch(); res = enc.results()
assert (res["status"] == 0).all()
if a.what in ("encode", "both"):
    print("encode ms/launch", enc.time(a.iters) / a.iters, "Mpx/s", a.images * a.size * a.size / (enc.time(a.iters) / a.iters) / 1e3)
if a.what in ("decode", "both"):
    dec = sq.DecodePlan(enc.out, res["len"], a.size, a.size, a.out_channels)
    dec.launch(); dres = dec.results()
    assert (dres["status"] == 0).all()
    if not os.environ.get("SQOA_PROF_NOCHECK"):   # (ablation builds -- stores or the serial part compiled out -- give wrong pixels on purpose)
        assert torch.equal(dec.out[..., :min(a.channels, a.out_channels)], px[..., :min(a.channels, a.out_channels)])
    print("decode ms/launch", dec.time(a.iters) / a.iters, "Mpx/s", a.images * a.size * a.size / (dec.time(a.iters) / a.iters) / 1e3)
