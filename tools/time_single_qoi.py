#!/usr/bin/env python3
"""One large QOI-compat image alone: strips (SQOA_HIP_STRIPS=1) against one workgroup (=0), encode and decode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for (w, h, n) in ((1024, 1024, 1), (3840, 2160, 1), (1024, 1024, 16), (1024, 1024, 64), (1024, 1024, 256), (1024, 1024, 1024)):
    px = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
    row = []
    for knob in ("0", "1", ""):
        os.environ["SQOA_HIP_STRIPS"] = knob
        enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results(); assert (res["status"] == 0).all()
        for _ in range(3): enc.time(5)
        row.append(min(enc.time(5) / 5 for _ in range(3)))
        enc.close()
    print(f"{n} x {w}x{h} QOI-compat encode: one workgroup {row[0]:.3f} ms, strips {row[1]:.3f} ms, plan's choice {row[2]:.3f} ms")
