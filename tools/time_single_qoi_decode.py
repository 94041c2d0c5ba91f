#!/usr/bin/env python3
"""Few large QOI-compat streams: the speculative chip-wide route (SQOA_HIP_MULTIPASS=1) against one workgroup per stream (=0)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for (w, h, n, cls) in ((1024, 1024, 1, "mixed"), (3840, 2160, 1, "mixed"), (1024, 1024, 16, "mixed"), (1024, 1024, 64, "mixed"), (1024, 1024, 256, "mixed"),
                       (3840, 2160, 1, "photo"), (2048, 2048, 1, "icon")):
    ch = 4 if cls == "mixed" else 3
    px = torch.empty((n, h, w, ch), dtype=torch.uint8, device="cuda"); sq.synth(px, cls, 1, 0)
    enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results(); assert (res["status"] == 0).all()
    row = []
    for knob in ("0", "1", ""):
        os.environ["SQOA_HIP_MULTIPASS"] = knob
        dec = sq.DecodePlan(enc.out, res["len"], w, h, ch); dec.launch(); r = dec.results()
        assert (r["status"] == 0).all() and torch.equal(dec.out, px), (knob, r["status"].tolist())
        for _ in range(2): dec.time(3)
        row.append(min(dec.time(3) / 3 for _ in range(2)))
        dec.close()
    print(f"{n} x {w}x{h} {cls}: QOI-compat decode one workgroup {row[0]:.3f} ms, speculative passes {row[1]:.3f} ms, plan's choice {row[2]:.3f} ms", flush=True)
