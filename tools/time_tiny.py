#!/usr/bin/env python3
"""Encode / decode time of 4096 small RGB icons (64^2, 128^2, 256^2), with a round-trip check: waves per image A/B via SQOA_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for size in (64, 128, 256):
    px = torch.empty((4096, size, size, 3), dtype=torch.uint8, device="cuda"); sq.synth(px, "icon", 1, 0)
    enc = sq.EncodePlan(px); enc.launch(); res = enc.results()
    assert (res["status"] == 0).all()
    dec = sq.DecodePlan(enc.out, res["len"], size, size, 3); dec.launch(); r = dec.results()
    assert (r["status"] == 0).all() and torch.equal(dec.out, px)
    enc.time(5); dec.time(5)
    e = min(enc.time(10) for _ in range(3)) / 10; d = min(dec.time(10) for _ in range(3)) / 10
    print(f"{os.environ.get('SQOA_LIB','default')} 4096 x {size}^2 RGB icons: encode {e*1e3:.1f} us, decode {d*1e3:.1f} us", flush=True)
    enc.close(); dec.close()
