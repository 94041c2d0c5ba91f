#!/usr/bin/env python3
"""Encode time of batches of FEW images: one workgroup per image against strips over the whole chip (SQOA_HIP_STRIPS)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for (w, h, ns) in ((1024, 1024, (1, 8, 64, 128, 256, 512, 1024)), (3840, 2160, (1, 4, 32))):
    for n in ns:
        px = torch.empty((n, h, w, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
        row, ref = [], None
        for knob in ("0", "1", ""):
            os.environ["SQOA_HIP_STRIPS"] = knob
            enc = sq.EncodePlan(px); enc.launch(); r = enc.results()
            assert (r["status"] == 0).all(), r["status"]
            lens = [int(x) for x in r["len"]]
            if ref is None: ref, ref_out = lens, enc.out.clone()
            else:
                assert lens == ref
                assert all(torch.equal(enc.out[i, : lens[i]], ref_out[i, : lens[i]]) for i in range(0, n, max(1, n // 8)))
            enc.time(5)
            row.append(min(enc.time(10) for _ in range(2)) / 10)
            enc.close()
        print(f"{n} x {w}x{h}: one workgroup per image {row[0]:.3f} ms, strips {row[1]:.3f} ms, the plan's own choice {row[2]:.3f} ms", flush=True)
