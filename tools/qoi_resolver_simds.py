#!/usr/bin/env python3
"""Where the resolver waves of decode_qoi_kernel sit: a diagnostic build (tools/mkvariant.sh NAME "-DSQOA_QDEC_HWDUMP [-DSQOA_QDEC_RSEL=..]") leaves
HW_ID of every stream's resolver wave in the result row's length; this prints, over the compute units, how many resolvers share a SIMD.
SQOA_LIB=variants/NAME.so python tools/qoi_resolver_simds.py [images]"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
os.environ.pop("SQOA_LIB_ENC", None)
enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, 4); dec.launch(); r = dec.results()
hw = [int(x) & 0xffffffff for x in r["len"]]
per_cu = collections.defaultdict(list)
for i, h in enumerate(hw):
    simd, cu, sh, se, tg = (h >> 4) & 3, (h >> 8) & 15, (h >> 12) & 1, (h >> 13) & 7, (h >> 16) & 15
    per_cu[(h >> 28, se, sh, cu)].append((simd, tg, i))
shape = collections.Counter()
for k, v in per_cu.items():
    c = collections.Counter(s for s, _, _ in v)
    shape[tuple(sorted(c.values(), reverse=True))] += 1
print(os.environ.get("SQOA_LIB", "default"), "compute units seen", len(per_cu), "resolvers per SIMD, by compute unit (sorted counts -> units):", dict(shape))
k0 = sorted(per_cu)[0]
print("one unit", k0, "(simd, tg_id, stream):", sorted(per_cu[k0]))
