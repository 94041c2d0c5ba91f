#!/usr/bin/env python3
"""One lone QOI-compat stream decoded a few times (for rocprofv3 --kernel-trace + tools/kernel_timeline.py): argv = w h class [images]."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
w, h, cls = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 1
ch = 4 if cls == "mixed" else 3
px = torch.empty((n, h, w, ch), dtype=torch.uint8, device="cuda"); sq.synth(px, cls, 1, 0)
enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], w, h, ch)
for _ in range(4): dec.launch()
torch.cuda.synchronize()
print(dec.results()["status"].tolist())
