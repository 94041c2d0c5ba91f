#!/usr/bin/env python3
"""Where a legacy sqoa_encode / sqoa_decode call on one 1024x1024 RGBA image spends its time: copies vs kernel."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, seqoia_amd as sq
from oracle import pyoracle as po
w = h = 1024
px = po.synth(w, h, 4, "mixed", 1, 0)
def t(fn, k=20):
    fn(); torch.cuda.synchronize(); a = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - a) / k * 1e3
d = torch.empty(4 << 20, dtype=torch.uint8, device="cuda")
hp = torch.from_numpy(px.copy()); hpin = hp.clone().pin_memory(); out_pg = torch.empty(4 << 20, dtype=torch.uint8); out_pin = torch.empty(4 << 20, dtype=torch.uint8).pin_memory()
print(f"H2D 4 MiB pageable {t(lambda: d.copy_(hp)):.3f} ms, pinned {t(lambda: d.copy_(hpin, non_blocking=True)):.3f} ms; D2H pageable {t(lambda: out_pg.copy_(d)):.3f} ms, pinned {t(lambda: out_pin.copy_(d, non_blocking=True)):.3f} ms; host memcpy 4 MiB {t(lambda: hpin.copy_(hp)):.3f} ms")
img = d.view(1, h, w, 4)
enc = sq.EncodePlan(img); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], w, h, 4); dec.launch(); dec.results()
print(f"one image alone: encode kernel {enc.time(20)/20:.3f} ms, decode kernel {dec.time(20)/20:.3f} ms")
desc = sq.Desc(w, h, 4, 0, 0)
s = sq.encode(px, desc)
print(f"legacy sqoa_encode {t(lambda: sq.encode(px, desc), 10):.3f} ms/call, sqoa_decode {t(lambda: sq.decode(s, 4), 10):.3f} ms/call (incl. the Python wrapper's own copies)")
