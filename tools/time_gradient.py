#!/usr/bin/env python3
"""Decode time of content made of ONE op kind: a horizontal ramp (every pixel one DIFF op: 1 byte per pixel), a ramp with step 5 (LUMA ops: 2 bytes),
uniform noise (RGB literals: 4 bytes).  Uniform op lengths are the balanced decoder's worst case twice over: 1024 one-byte ops in a chunk are more
than its 512 op slots (the stream goes to the stand-by kernels), and equal strides between the lanes' ops are LDS bank conflicts.
SQOA_LIB / SQOA_HIP_DEC_BAL select the library and the plan's kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
n, s = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 1024
x = torch.arange(s, device="cuda", dtype=torch.int32)
for name, step in (("diff", 1), ("luma", 5), ("literal", 0)):
    if step:
        row = ((x * step) & 255).to(torch.uint8)
        px = torch.stack([row, row, row, torch.full_like(row, 255)], dim=-1).expand(n, s, s, 4).contiguous()
    else:
        px = torch.randint(0, 256, (n, s, s, 4), device="cuda", dtype=torch.uint8); px[..., 3] = 255
    enc = sq.EncodePlan(px, qoi_compat=0); enc.launch(); res = enc.results()
    dec = sq.DecodePlan(enc.out, res["len"], s, s, 4); dec.launch(); r = dec.results()
    assert (r["status"] == 0).all() and torch.equal(dec.out, px)
    for _ in range(3): dec.time(3)
    ms = min(dec.time(3) / 3 for _ in range(3))
    print(f"{os.environ.get('SQOA_LIB','default')} BAL={os.environ.get('SQOA_HIP_DEC_BAL','-')} {name:8s} rate {float(res['len'].sum()) / px.numel():.3f}  decode {ms:.3f} ms  {n * s * s / ms / 1e3:.0f} Mpx/s", flush=True)
    dec.close(); enc.close()
