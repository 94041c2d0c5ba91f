import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
for cls, n, sz in (("icon", 4096, 512), ("photo", 1024, 1024)):
    px = torch.empty((n, sz, sz, 3), dtype=torch.uint8, device="cuda")
    sq.synth(px, cls, 1, 0)
    enc = sq.EncodePlan(px, qoi_compat=0); enc.launch(); res = enc.results()
    for oc in (3, 4):
        dec = sq.DecodePlan(enc.out, res["len"], sz, sz, oc); dec.launch(); r = dec.results()
        assert (r["status"] == 0).all()
        if oc == 3: assert torch.equal(dec.out, px)
        ms = dec.time(3) / 3
        print(cls, "out_channels", oc, f"{ms:.3f} ms  {n*sz*sz/ms/1e3:.0f} Mpx/s")
        dec.close()
    ms = enc.time(3)/3
    print(cls, "encode rgb", f"{ms:.3f} ms {n*sz*sz/ms/1e3:.0f} Mpx/s")
