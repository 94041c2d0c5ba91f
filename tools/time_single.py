#!/usr/bin/env python3
"""Kernel time of ONE image alone on the device (what the legacy per-image entry points and the largest image of a mixed batch see)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for (w, h) in ((1024, 1024), (3840, 2160)):
    px = torch.empty((1, h, w, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
    enc = sq.EncodePlan(px); enc.launch(); res = enc.results()
    dec = sq.DecodePlan(enc.out, res["len"], w, h, 4); dec.launch(); dec.results()
    enc.time(5); dec.time(5)
    print(f"{os.environ.get('SQOA_LIB','default')} {w}x{h}: encode {enc.time(10)/10:.3f} ms, decode {dec.time(10)/10:.3f} ms")
