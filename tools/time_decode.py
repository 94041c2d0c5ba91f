#!/usr/bin/env python3
"""Time the decode plan only (no output check): for A/B runs of experimental builds via SQOA_LIB.
The streams are produced by the default library's encoder so that every variant decodes the same bytes."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seqoia_amd as sq
ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=1024); ap.add_argument("--size", type=int, default=1024)
ap.add_argument("--iters", type=int, default=5); ap.add_argument("--cls", default="mixed")
ap.add_argument("--flat", action="store_true", help="(default layout; kept for old command lines)")
ap.add_argument("--channels", type=int, default=4); ap.add_argument("--out-channels", type=int, default=4)
ap.add_argument("--qoi", type=int, default=0, help="1: QOI-compat streams")
a = ap.parse_args()
px = torch.empty((a.images, a.size, a.size, a.channels), dtype=torch.uint8, device="cuda")
sq.synth(px, a.cls, 1, 0)
enc = sq.EncodePlan(px, qoi_compat=a.qoi)
enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], a.size, a.size, a.out_channels)
dec.launch(); dec.results()
for _ in range(4): dec.time(a.iters)  # the first ~50 ms after an idle period run at lower clocks (tools/time_warm.py)
ms = min(dec.time(a.iters) / a.iters for _ in range(3))
print(f"{os.environ.get('SQOA_LIB','default')} cls={a.cls} qoi={a.qoi} {a.channels} -> {a.out_channels} channels decode {ms:.3f} ms/launch  {a.images*a.size*a.size/ms/1e3:.0f} Mpx/s")
