"""Which classes / images does the speculative QOI-compat decode route hold on?  (A stream it had to hand back shows as a decode time of a
one-workgroup decode, > 1 ms for 1 Mpixel: a timing heuristic -- tools/qspec_why.py reads the check's own verdict from a -DSQOA_DQ_DEBUG build.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
os.environ["SQOA_HIP_MULTIPASS"] = "1"
for cls, ch in (("mixed", 4), ("photo", 3), ("icon", 3), ("mixed", 3)):
    fails = []
    for img in range(24):
        px = torch.empty((1, 1024, 1024, ch), dtype=torch.uint8, device="cuda"); sq.synth(px, cls, 1, img)
        enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
        dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, ch); dec.launch(); dec.results()
        t = min(dec.time(2) / 2 for _ in range(2))
        if t > 1.0: fails.append((img, round(t, 2)))
        dec.close(); enc.close()
    print(cls, ch, "speculation failed for", fails)
