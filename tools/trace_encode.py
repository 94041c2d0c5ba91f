#!/usr/bin/env python3
"""Workgroup timeline of the encode kernel (needs a -DSQOA_TRACE build selected with SQOA_LIB)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import seqoia_amd as sq
ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=1024); ap.add_argument("--size", type=int, default=1024)
a = ap.parse_args()
px = torch.empty((a.images, a.size, a.size, 4), dtype=torch.uint8, device="cuda")
sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px)
for _ in range(3): enc.launch()
torch.cuda.synchronize(); enc.results()
off = ((enc.out.data_ptr() + enc.slot - 32) & ~7) - enc.out.data_ptr()
raw = enc.out[:, off:off + 24].cpu().numpy().copy().view(np.uint64).reshape(a.images, 3)
t0 = raw[:, 0].astype(np.int64); t1 = raw[:, 1].astype(np.int64); hw = raw[:, 2]
base = t0.min(); s = (t0 - base) / 100.0; e = (t1 - base) / 100.0  # us (100 MHz)
d = e - s
print(f"n={a.images} span {e.max():.0f} us; start pct [0,50,75,90,100] {np.percentile(s,[0,50,75,90,100]).round(0)}")
print(f"end time us: p1 {np.percentile(e, 1):.0f} median {np.median(e):.0f} p90 {np.percentile(e, 90):.0f} p99 {np.percentile(e, 99):.0f} max {e.max():.0f}  (tail behind the median: {100 * (e.max() - np.median(e)) / e.max():.1f} % of the span)")
print(f"duration us: min {d.min():.0f} p10 {np.percentile(d,10):.0f} p50 {np.percentile(d,50):.0f} p90 {np.percentile(d,90):.0f} max {d.max():.0f}")
late = s > 100
print(f"first-round WGs {np.sum(~late)}: dur mean {d[~late].mean():.0f}; later WGs {np.sum(late)}: dur mean {d[late].mean() if late.any() else 0:.0f}, start mean {s[late].mean() if late.any() else 0:.0f}")
xcc = (hw >> np.uint64(32)) & np.uint64(0xf); cu = (hw >> np.uint64(8)) & np.uint64(0xf); se = (hw >> np.uint64(13)) & np.uint64(0x7); sh = (hw >> np.uint64(12)) & np.uint64(1)
key = xcc * np.uint64(1000) + se * np.uint64(100) + sh * np.uint64(50) + cu
u, c = np.unique(key[~late], return_counts=True)
print(f"first round: {len(u)} distinct (xcc,se,sh,cu), WGs per CU histogram {np.bincount(c)}")
# concurrency over time
ts = np.linspace(0, e.max(), 21)
print("t_us:resident ", " ".join(f"{t:.0f}:{int(np.sum((s <= t) & (e > t)))}" for t in ts))
lens = enc.results()["len"]
print("corr(duration, len) =", np.corrcoef(d, lens)[0, 1].round(3))
ukey, inv = np.unique(key.astype(np.int64), return_inverse=True)
cu_end = np.array([e[inv == k].max() for k in range(len(ukey))]); cu_len = np.array([lens[inv == k].sum() for k in range(len(ukey))])
print(f"CU end time us: median {np.median(cu_end):.0f} p99 {np.percentile(cu_end, 99):.0f} max {cu_end.max():.0f}; encoded bytes per CU max / mean {cu_len.max() / cu_len.mean():.4f}; corr(CU end, CU bytes) {np.corrcoef(cu_end, cu_len)[0, 1]:.3f}")

ph = enc.out[:, off - 64:off].cpu().numpy().copy().view(np.uint64).reshape(a.images, 8).astype(np.float64)
tot = ph.sum(1, keepdims=True)
names = ["fetch issue", "classify", "run chain", "emit", "take (wait next px)", "pos chain wait", "flush", "loop glue"]
print("wave 0 time by phase (mean % over images):", ", ".join(f"{n} {v:.1f}" for n, v in zip(names, (100 * ph / tot).mean(0))))
