#!/usr/bin/env python3
"""How the QOI-compat resolver settled its groups (a -DSQOA_QCOUNT build, SQOA_LIB=variants/qc.so): per image the number of
groups with INDEX ops that went through the slot-first path, how many of them needed its second attempt, and how many fell
through to the fixed point.  (The counters overwrite the first three dwords of every decoded image.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for name, shape, cls in (("headline 1024 x 1 Mpixel RGBA mixed", (256, 1024, 1024, 4), "mixed"), ("icons 512^2 RGB", (1024, 512, 512, 3), "icon"),
                         ("photo 1 Mpixel RGB", (128, 1024, 1024, 3), "photo"), ("icons 512^2 RGBA", (1024, 512, 512, 4), "icon")):
    px = torch.empty(shape, dtype=torch.uint8, device="cuda"); sq.synth(px, cls, 1, 0)
    enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
    dec = sq.DecodePlan(enc.out, res["len"], shape[1], shape[2], 4); dec.launch(); dec.results()
    out = dec.out.view(torch.int32).view(shape[0], -1)[:, :3].cpu().long().sum(0)
    print(f"{name}: groups with INDEX ops {int(out[0])}, second attempts {int(out[1])} ({100.0 * int(out[1]) / max(1, int(out[0])):.2f} %), fixed point {int(out[2])} ({100.0 * int(out[2]) / max(1, int(out[0])):.2f} %)")
