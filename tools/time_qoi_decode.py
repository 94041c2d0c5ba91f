#!/usr/bin/env python3
"""QOI-compat decode time of the headline-shaped batch (streams from the default library's encoder)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, 4); dec.launch(); dec.results()
dec.time(1)
print(os.environ.get("SQOA_LIB", "default"), n, "images: qoi decode", round(min(dec.time(1) for _ in range(3)), 2), "ms")
