#!/usr/bin/env python3
"""Kernel times (HIP events, warm) for the BASELINE.json configurations other than the headline batch:
QOI-compat, 3-channel icons (C4 sizes), photo class, plus the headline for reference."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
def warm(plan, k=3):
    for _ in range(3): plan.time(k)
    return min(plan.time(k) / k for _ in range(3))
rows = [("mixed", 1024, 1024, 4, 0), ("mixed", 1024, 1024, 4, 1), ("photo", 1024, 1024, 3, 0), ("icon", 4096, 512, 3, 0), ("icon", 4096, 256, 3, 0),
        ("icon", 4096, 128, 3, 0), ("icon", 4096, 64, 3, 0), ("icon", 4096, 512, 3, 1)]
for cls, n, sz, ch, qoi in rows:
    px = torch.empty((n, sz, sz, ch), dtype=torch.uint8, device="cuda")
    sq.synth(px, cls, 1, 0)
    enc = sq.EncodePlan(px, qoi_compat=qoi); enc.launch(); res = enc.results()
    assert (res["status"] == 0).all()
    e_ms = warm(enc)
    dec = sq.DecodePlan(enc.out, res["len"], sz, sz, ch); dec.launch(); r = dec.results()
    assert (r["status"] == 0).all() and torch.equal(dec.out, px)
    d_ms = warm(dec, 1 if qoi else 3)
    mp = n * sz * sz / 1e3
    print(f"{cls:5s} {n:5d} x {sz:4d}^2 x{ch} qoi_compat={qoi}: rate {res['len'].sum() / (n * sz * sz * ch):.3f}  encode {e_ms:8.3f} ms {mp / e_ms:9.0f} Mpx/s   decode {d_ms:8.3f} ms {mp / d_ms:9.0f} Mpx/s", flush=True)
    enc.close(); dec.close(); del px, enc, dec
