#!/usr/bin/env python3
"""Encode time of the plans that go through strips (the plan's own choice): one 1024x1024 image, 16 of them, one 3840x2160 image; SQOA and
QOI-compat.  For A/B runs via SQOA_LIB."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
out = []
for (w, h, n) in ((1024, 1024, 1), (1024, 1024, 16), (3840, 2160, 1)):
    px = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
    for q in (0, 1):
        enc = sq.EncodePlan(px, qoi_compat=q); enc.launch(); r = enc.results()
        assert (r["status"] == 0).all()
        enc.time(20)
        out.append(f"{n} x {w}x{h} q{q}: {min(enc.time(20) for _ in range(3)) / 20 * 1000:.1f} us")
        enc.close()
print(os.environ.get("SQOA_LIB", "default"), "; ".join(out))
