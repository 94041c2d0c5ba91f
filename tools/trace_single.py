#!/usr/bin/env python3
"""One 1024x1024 image alone, encode then decode, a few launches each: run under rocprofv3 --kernel-trace to see the
launches a lone image costs (kernel durations and the gaps between them).  usage: trace_single.py [w h] [qoi]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
w, h = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1024, 1024)
qoi = int(sys.argv[3]) if len(sys.argv) > 3 else 0
px = torch.empty((1, h, w, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px, qoi_compat=qoi); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], w, h, 4); dec.launch(); dec.results()
for _ in range(3):
    print("encode %.4f ms  decode %.4f ms" % (enc.time(20) / 20, dec.time(20) / 20))
