#!/usr/bin/env python3
"""Where the QOI-compat resolver's cycles go (a -DSQOA_QTRACE build, SQOA_LIB=variants/qt.so): s_memtime deltas per phase, summed
by the resolver wave over its image and written over the image's first pixels (the first chunk's producer writes pixel 0.. again:
images whose counters did not survive are skipped).  Per chunk and per group of 64 records."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq, numpy as np
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
cls = sys.argv[2] if len(sys.argv) > 2 else "mixed"
px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, cls, 1, 0)
enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, 4); dec.launch(); dec.results()
print(os.environ.get("SQOA_LIB", "default"), cls, "ms", round(dec.time(3) / 3, 3))
torch.cuda.synchronize()
tr = dec.out.view(n, -1)[:, :96].cpu().numpy().view(np.uint64).astype(np.float64)
ok = (tr[:, 10] > 1000) & (tr[:, 10] < 20000) & (tr[:, 8] > 0)
tr = tr[ok]
names = ["top of chunk", "wait for the producer", "records in, decoded", "INDEX groups: heads, index look-up issued", "pixels settled", "behind the last group", "index update, hand-back"]
ch, gr = tr[:, 10].sum(), tr[:, 8].sum()
print(f"{len(tr)} images: {ch / len(tr):.0f} chunks, {gr / ch:.2f} groups per chunk, {100 * tr[:, 9].sum() / gr:.1f} % of the groups with INDEX ops")
for k, nm in enumerate(names):
    print(f"  {nm:28s} {tr[:, k].sum() / ch:8.0f} cycles per chunk  {tr[:, k].sum() / gr:7.0f} per group")
print(f"  {'sum':28s} {tr[:, :7].sum() / ch:8.0f}")
