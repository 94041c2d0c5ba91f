#!/usr/bin/env python3
"""Where wave 0 of every decode workgroup spends its time (needs a -DSQOA_TRACE build selected with SQOA_LIB)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seqoia_amd as sq
n, s, gap = 1024, 1024, 4096
px = torch.empty((n, s, s, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px); enc.launch(); lens = enc.results()["len"]
img = s * s * 4
arena = torch.zeros(n * (img + gap), dtype=torch.uint8, device="cuda")  # the trace lands in the gap behind each image
out = arena.as_strided((n, s, s, 4), (img + gap, s * 4, 4, 1))
dec = sq.DecodePlan(enc.out, lens, s, s, 4, out=out)
for _ in range(6): dec.launch()
torch.cuda.synchronize(); assert (dec.results()["status"] == 0).all()
raw = arena.as_strided((n, 88), (img + gap, 1), img).cpu().numpy().copy().view(np.uint64).reshape(n, 11)
ph = raw[:, :8].astype(np.float64)
t0, t1 = raw[:, 8].astype(np.int64), raw[:, 9].astype(np.int64)
s0, e0 = (t0 - t0.min()) / 100.0, (t1 - t0.min()) / 100.0   # us (100 MHz)
print(f"decode workgroups: span {e0.max():.0f} us; start pct [0,50,100] {np.percentile(s0, [0, 50, 100]).round(1)}; end time us: p1 {np.percentile(e0, 1):.0f} median {np.median(e0):.0f} "
      f"p90 {np.percentile(e0, 90):.0f} p99 {np.percentile(e0, 99):.0f} max {e0.max():.0f}  (tail behind the median: {100 * (e0.max() - np.median(e0)) / e0.max():.1f} % of the span)")
hwid = raw[:, 10]; xcc = (hwid >> np.uint64(32)) & np.uint64(0xf); cu = (hwid >> np.uint64(8)) & np.uint64(0xf); se = (hwid >> np.uint64(13)) & np.uint64(0x7); sh = (hwid >> np.uint64(12)) & np.uint64(1)
key = (xcc * np.uint64(1000) + se * np.uint64(100) + sh * np.uint64(50) + cu).astype(np.int64)
u, inv = np.unique(key, return_inverse=True)
cu_end = np.array([e0[inv == k].max() for k in range(len(u))]); cu_bytes = np.array([lens[inv == k].sum() for k in range(len(u))])
print(f"{len(u)} compute units, workgroups per CU {np.bincount(np.bincount(inv))}; CU end time us: median {np.median(cu_end):.0f} p99 {np.percentile(cu_end, 99):.0f} max {cu_end.max():.0f}; "
      f"stream bytes per CU max / mean {cu_bytes.max() / cu_bytes.mean():.4f}; corr(CU end, CU bytes) {np.corrcoef(cu_end, cu_bytes)[0, 1]:.3f}")
# (decode_wave_kernel marks parse / parse chain / walk / scans / chain wait / stores; the balanced kernel parse / parse chain / starts / walk / scans + chain wait / stores)
bal = os.environ.get("SQOA_HIP_DEC_BAL", "") != "0"
names = (["parse", "parse chain", "starts", "walk", "scans + chain wait", "store", "-", "loop glue"] if bal else
         ["stage+parse", "parse chain", "walk", "scans", "pos/px chain wait", "store", "-", "loop glue"])
print("decode wave 0 time by phase (mean % over images):", ", ".join(f"{a} {v:.1f}" for a, v in zip(names, (100 * ph / ph.sum(1, keepdims=True)).mean(0))))
