"""Why the speculative QOI-compat decode route rejects a stream: first failing workgroup and reason code, as left in the result row by
dq_verify_kernel -- NEEDS a build with -DSQOA_DQ_DEBUG (tools/mkvariant.sh dqdbg "-DSQOA_DQ_DEBUG", SQOA_LIB=variants/dqdbg.so); a plain
build reports nothing here."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, seqoia_amd as sq
os.environ["SQOA_HIP_MULTIPASS"] = "1"
for cls, ch in (("icon", 3), ("mixed", 3), ("photo", 3), ("mixed", 4), ("photo", 4)):
    for img in range(4):
        px = torch.empty((1, 1024, 1024, ch), dtype=torch.uint8, device="cuda"); sq.synth(px, cls, 1, img)
        enc = sq.EncodePlan(px, qoi_compat=1); enc.launch(); res = enc.results()
        dec = sq.DecodePlan(enc.out, res["len"], 1024, 1024, ch); dec.launch(); r = dec.results()
        code = int(r["len"][0]); st = int(r["status"][0])
        print(cls, ch, img, "status", st, "first failing workgroup", code >> 4 if st == 1 else None, "of", (int(res["len"][0]) - 22 + 511) // 512, "chunks; flags (1: a wave did not finish, 2: exposed read not known good, 4: exposed read differs outright, 8: pixel differs)", code & 15 if st == 1 else None)
        dec.close(); enc.close()
