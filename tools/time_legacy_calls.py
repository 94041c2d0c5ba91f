import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch, seqoia_amd as sq
from oracle import pyoracle as po
w = h = 1024
px = po.synth(w, h, 4, "mixed", 1, 0)
desc = sq.Desc(w, h, 4, 0, 0)
s = sq.encode(px, desc)
for k in range(8):
    a = time.perf_counter(); out = sq.decode(s, 4); b = time.perf_counter()
    print(f"call {k}: {(b-a)*1e3:.3f} ms", flush=True)
