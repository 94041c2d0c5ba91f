#!/usr/bin/env python3
"""The encode -> decode step with an event between the two plan launches: which of the two is slower inside the step than alone?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, seqoia_amd as sq
n = 1024
px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, 4); dec.launch(); dec.results()
st = torch.cuda.Stream(); s = st.cuda_stream
def spin(secs):
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < secs:
        for _ in range(8): enc.launch(s); dec.launch(s)
        torch.cuda.synchronize()
for rep in range(4):
    spin(1.0)
    K = 60
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    with torch.cuda.stream(st):
        for k in range(K):
            ev[k][0].record(); enc.launch(s); ev[k][1].record(); dec.launch(s); ev[k][2].record()
    torch.cuda.synchronize()
    e = np.array([ev[k][0].elapsed_time(ev[k][1]) for k in range(K)]); d = np.array([ev[k][1].elapsed_time(ev[k][2]) for k in range(K)])
    tot = ev[0][0].elapsed_time(ev[K - 1][2]) / K
    print(f"in the step: encode {e.mean():.3f} (min {e.min():.3f} max {e.max():.3f}) decode {d.mean():.3f} (min {d.min():.3f} max {d.max():.3f}); step {tot:.3f} ms", flush=True)
    a = enc.time(20) / 20; b = dec.time(20) / 20
    print(f"alone right after: encode {a:.3f} decode {b:.3f} sum {a + b:.3f}", flush=True)
