#!/usr/bin/env python3
"""tools/pmc_latest.py SUMMARY.txt TAG -> profiles/pmc_hbm_latest.json: the HBM counters of the headline kernels from a
tools/pmc_table.py summary, stamped with the hash of seqoia_amd/csrc so that bench.py reports `roofline.traffic` only
for the kernel source the counters were taken from."""
import hashlib, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "seqoia_amd", "csrc")
h = hashlib.sha256()
for f in sorted(os.listdir(src)):
    h.update(open(os.path.join(src, f), "rb").read())
kern, cur = {}, None
for ln in open(sys.argv[1]):
    if ln.startswith("sqoa::"):
        cur = ln.strip()
        kern[cur] = {}
    elif cur and ln.strip():
        m = re.match(r"\s+(\S+)\s+([0-9.]+)", ln)
        if m:
            kern[cur][m.group(1)] = float(m.group(2))
out_images = 1024
pick = lambda pre: next((v for k, v in kern.items() if k.startswith(pre)), None)
enc, dec = pick("sqoa::encode_kernel<8, false>"), (pick("sqoa::decode_wave_kernel<8, 0>") or pick("sqoa::decode_wave_kernel"))
out = {"note": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) of bench.py on the headline batch (1024 x 1024x1024 RGBA), KB per launch; "
               f"gfx950: FETCH_SIZE counts 1/2 of streamed read bytes (MI355X_MICROARCH.md, HBM) -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024. "
               f"Full table: profiles/{sys.argv[2]}_pmc_summary.txt (tools/profile_round.sh {sys.argv[2]})",
       "csrc_sha256": h.hexdigest(), "images": out_images, "size": 1024, "kernels": {}}
for name, k in (("encode", enc), ("decode", dec)):
    if k and "FETCH_SIZE" in k and "WRITE_SIZE" in k:
        out["kernels"][name] = {"FETCH_SIZE_KB": k["FETCH_SIZE"], "WRITE_SIZE_KB": k["WRITE_SIZE"]}
json.dump(out, open(os.path.join(root, "profiles", "pmc_hbm_latest.json"), "w"), indent=1)
print(json.dumps(out["kernels"]))
