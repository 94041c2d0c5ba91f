import os, sys
sys.path.insert(0, "/root/repo")
import torch, seqoia_amd as sq, numpy as np
n = 1024
px = torch.empty((n, 1024, 1024, 4), dtype=torch.uint8, device="cuda"); sq.synth(px, "mixed", 1, 0)
enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, 4); dec.launch(); dec.results()
print("ms", dec.time(3)/3)
torch.cuda.synchronize()
# note: the producers overwrite the first pixels again; read what survived for several images
for i in (0, 100, 500, 1000):
    tr = dec.out[i].view(-1)[:64].cpu().numpy().view(np.uint64)
    print(i, [int(x) for x in tr])
