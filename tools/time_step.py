#!/usr/bin/env python3
"""Encode, decode and the encode->decode step of the headline batch timed in one process (HIP events on one stream)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
n = 1024
px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, 4); dec.launch(); dec.results()
st = torch.cuda.Stream(); s = st.cuda_stream
def timed(fn, k=20):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(st):
        a.record()
        for _ in range(k): fn()
        b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / k
for rep in range(2):
    e = timed(lambda: enc.launch(s)); d = timed(lambda: dec.launch(s))
    both = timed(lambda: (enc.launch(s), dec.launch(s)))
    print(f"encode {e:.3f} ms, decode {d:.3f} ms, sum {e+d:.3f}; step {both:.3f} ms")
import time
def wall(fn, k=20):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / k * 1e3
for rep in range(3):
    print(f"wall-clock step {wall(lambda: (enc.launch(s), dec.launch(s))):.3f} ms; host time to enqueue one step:", end=" ")
    t0 = time.perf_counter()
    for _ in range(20): enc.launch(s); dec.launch(s)
    t1 = time.perf_counter(); torch.cuda.synchronize()
    print(f"{(t1 - t0) / 20 * 1e3:.3f} ms")
for k in (5, 20, 100, 400):
    print(f"k={k}: wall-clock step {wall(lambda: (enc.launch(s), dec.launch(s)), k):.3f} ms")
t0 = time.perf_counter(); torch.cuda.synchronize(); print(f"idle synchronize: {(time.perf_counter()-t0)*1e3:.3f} ms")
