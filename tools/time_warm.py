#!/usr/bin/env python3
"""Does the measured kernel time depend on how long the GPU has been busy?  Repeats enc.time() back to back."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import seqoia_amd as sq
ap = argparse.ArgumentParser(); ap.add_argument("--reps", type=int, default=12); ap.add_argument("--iters", type=int, default=5)
a = ap.parse_args()
n, s = 1024, 1024
px = torch.empty((n, s, s, 4), dtype=torch.uint8, device="cuda")
sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], s, s, 4); dec.launch(); dec.results()
out = []
for r in range(a.reps):
    out.append((enc.time(a.iters) / a.iters, dec.time(a.iters) / a.iters))
print("encode/decode ms per launch, back to back:", " ".join(f"{e:.3f}/{d:.3f}" for e, d in out))
