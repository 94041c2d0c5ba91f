#!/usr/bin/env python3
"""RGB streams decoded to 3 channels with the wave decoder at 8 and at 12 stream bytes per lane (SQOA_HIP_DEC_WIDE=0 / 1): photo 1024 x 1 Mpixel, icons 4096 x 512 x 512."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for cls, n, sz in (("photo", 1024, 1024), ("icon", 4096, 512)):
    px = torch.empty((n, sz, sz, 3), dtype=torch.uint8, device="cuda"); sq.synth(px, cls, 1, 0)
    enc = sq.EncodePlan(px, qoi_compat=0); enc.launch(); res = enc.results()
    for wide in ("0", "1"):
        os.environ["SQOA_HIP_DEC_WIDE"] = wide
        dec = sq.DecodePlan(enc.out, res["len"], sz, sz, 3); dec.launch(); r = dec.results()
        assert (r["status"] == 0).all() and torch.equal(dec.out, px)
        for _ in range(3): dec.time(3)
        ms = min(dec.time(3) / 3 for _ in range(3))
        print(os.environ.get("SQOA_LIB", "default"), cls, "to 3 channels,", "12" if wide == "1" else "8", f"bytes per lane: {ms:.3f} ms  {n*sz*sz/ms/1e3:.0f} Mpx/s", flush=True)
        dec.close()
    enc.close()
