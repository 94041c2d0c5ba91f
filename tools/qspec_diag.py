import os, sys
sys.path.insert(0, "/root/repo")
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
