#!/usr/bin/env python3
"""Decode time of a batch of mono+alpha images (record pipeline of decode_fallback_kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, seqoia_amd as sq
from seqoia_amd.api import Plan, DECODE, ENCODE, Job
n, w, h, c = 1024, 512, 512, 2
rng = np.random.default_rng(1)
raw = ((np.arange(w * h * c) // 9) + rng.integers(0, 2, w * h * c) * rng.integers(0, 40, w * h * c)).astype(np.uint8)
px = torch.from_numpy(np.tile(raw, (n, 1))).cuda()
d = sq.Desc(w, h, c, 0, 0)
slot = int(sq.encode_bound(d))
slots = torch.zeros((n, slot), dtype=torch.uint8, device="cuda")
enc = Plan(ENCODE, [Job(px.data_ptr() + i * w * h * c, slots.data_ptr() + i * slot, 0, slot, d, 0) for i in range(n)], 0)
enc.launch(); res = enc.results(); assert (res["status"] == 0).all()
e_ms = min(enc.time(3) / 3 for _ in range(3))
out = torch.zeros((n, w * h * c), dtype=torch.uint8, device="cuda")
dec = Plan(DECODE, [Job(slots.data_ptr() + i * slot, out.data_ptr() + i * w * h * c, int(res["len"][i]), w * h * c, sq.Desc(0, 0, 0, 0, 0), 0) for i in range(n)], 0)
dec.launch(); r = dec.results(); assert (r["status"] == 0).all() and torch.equal(out, px)
dec.time(1)
d_ms = min(dec.time(1) for _ in range(3))
print(f"{n} x {w}x{h} mono+alpha: rate {res['len'].sum() / (n * w * h * c):.3f} encode {e_ms:.3f} ms {n*w*h/e_ms/1e3:.0f} Mpx/s  decode {d_ms:.3f} ms {n*w*h/d_ms/1e3:.0f} Mpx/s")
