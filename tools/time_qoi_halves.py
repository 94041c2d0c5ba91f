import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for (n, h) in ((1024, 1024), (2048, 512), (3072, 341)):
    px = torch.empty((n, h, 1024, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
    enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
    dec = sq.DecodePlan(enc.out, res["len"], 1024, h, 4); dec.launch(); dec.results()
    dec.time(1)
    print(os.environ.get("SQOA_LIB", "default"), n, "x", h, "rows: qoi decode", round(min(dec.time(1) for _ in range(3)), 2), "ms", flush=True)
    dec.close(); enc.close(); del px
