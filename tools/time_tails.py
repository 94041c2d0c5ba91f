#!/usr/bin/env python3
"""What the two kernels behind the SQOA wave decoder cost a plan that does not need them: the same decode plan with and
without SQOA_HIP_JOB_WAVEONLY on its jobs (one image alone, and many small ones)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
from seqoia_amd import Desc, Job
from seqoia_amd.api import Plan, DECODE
for (w, h, n, cls, ch) in ((1024, 1024, 1, "mixed", 4), (3840, 2160, 1, "mixed", 4), (64, 64, 4096, "icon", 3), (1024, 1024, 16, "mixed", 4)):
    px = torch.empty((n, h, w, ch), dtype=torch.uint8, device="cuda"); sq.synth(px, cls, 1, 0)
    enc = sq.EncodePlan(px); enc.launch(); res = enc.results()
    out = torch.empty((n, h, w, ch), dtype=torch.uint8, device="cuda")
    row = []
    for flag in (0, 8):
        jobs = [Job(enc.out[i].data_ptr(), out[i].data_ptr(), int(res["len"][i]), w * h * ch, Desc(w, h, ch, 0, 0), flag) for i in range(n)]
        p = Plan(DECODE, jobs); p.launch(); assert (p.results()["status"] == 0).all() and torch.equal(out, px)
        for _ in range(3): p.time(20)
        row.append(min(p.time(20) / 20 for _ in range(3))); p.close()
    print(f"{n} x {w}x{h}: decode {row[0]*1e3:.1f} us with the QOI-compat / fallback launches, {row[1]*1e3:.1f} us without")
