#!/usr/bin/env python3
"""QOI-compat decode time of the RGB icon set 4096 x 512^2 (BASELINE.json configs[3])."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
px = torch.empty((4096, 512, 512, 3), dtype=torch.uint8, device="cuda"); sq.synth(px, "icon", 1, 0)
enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], 512, 512, 3); dec.launch(); dec.results()
dec.time(1)
print(os.environ.get("SQOA_LIB", "default"), "icons 4096 x 512^2 RGB: qoi decode", round(min(dec.time(1) for _ in range(3)), 2), "ms")
