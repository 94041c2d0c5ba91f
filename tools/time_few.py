#!/usr/bin/env python3
"""Decode time of batches of FEW 1024x1024 images, chained (one workgroup per stream) against three-pass (SQOA_HIP_MULTIPASS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for n in (1, 8, 64, 200, 256, 320, 512, 1024):
    px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", n, 0)
    enc = sq.EncodePlan(px); enc.launch(); res = enc.results()
    row = []
    for knob in ("0", "1", ""):
        os.environ["SQOA_HIP_MULTIPASS"] = knob
        dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, 4); dec.launch(); r = dec.results()
        assert (r["status"] == 0).all(), r["status"]
        assert torch.equal(dec.out, px)
        dec.time(5)
        row.append(dec.time(10) / 10)
    print(f"n={n}: chained {row[0]:.3f} ms, three-pass {row[1]:.3f} ms, the plan's own choice {row[2]:.3f} ms", flush=True)
