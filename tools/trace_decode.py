#!/usr/bin/env python3
"""Where wave 0 of every decode workgroup spends its time (needs a -DSQOA_TRACE build selected with SQOA_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seqoia_amd as sq
n, s, gap = 1024, 1024, 4096
px = torch.empty((n, s, s, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px); enc.launch(); lens = enc.results()["len"]
img = s * s * 4
arena = torch.zeros(n * (img + gap), dtype=torch.uint8, device="cuda")  # the trace lands in the gap behind each image
out = arena.as_strided((n, s, s, 4), (img + gap, s * 4, 4, 1))
dec = sq.DecodePlan(enc.out, lens, s, s, 4, out=out)
for _ in range(6): dec.launch()
torch.cuda.synchronize(); assert (dec.results()["status"] == 0).all()
ph = arena.as_strided((n, 64), (img + gap, 1), img).cpu().numpy().copy().view(np.uint64).reshape(n, 8).astype(np.float64)
names = ["stage+parse", "parse chain", "walk", "scans", "pos/px chain wait", "store", "-", "loop glue"]
print("decode wave 0 time by phase (mean % over images):", ", ".join(f"{a} {v:.1f}" for a, v in zip(names, (100 * ph / ph.sum(1, keepdims=True)).mean(0))))
