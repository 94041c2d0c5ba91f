"""Decode time of a 1 024 x 1 Mpixel RGBA list of mixed CONTENT -- N photographs (class photo) among screenshots (class mixed) -- with the plan's
content feedback (default) and without (SQOA_HIP_DEC_CONTENT=0): what the per-job split of sqoa_hip_plan_results is worth.  usage: time_mixed_content.py [N]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import seqoia_amd as sq
n, s = 1024, 1024
npho = int(sys.argv[1]) if len(sys.argv) > 1 else 256
px = torch.empty((n, s, s, 4), dtype=torch.uint8, device="cuda")
sq.synth(px[:npho], "photo", 1, 0); sq.synth(px[npho:], "mixed", 1, 0)
enc = sq.EncodePlan(px); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], s, s, 4); dec.launch(); dec.results()
for _ in range(4): dec.time(5)
ms = min(dec.time(5) / 5 for _ in range(3))
print(f"CONTENT={os.environ.get('SQOA_HIP_DEC_CONTENT','1')} {npho} photo + {n-npho} mixed: shape {hex(dec.shape())} decode {ms:.3f} ms")
