#!/usr/bin/env python3
"""tools/pmc_latest.py SUMMARY.txt TAG [KERNEL_STATS.csv] [KERNEL_PHASES.json] -> profiles/pmc_hbm_latest.json: the counters of the headline
kernels from a tools/pmc_table.py summary -- HBM bytes (FETCH_SIZE / WRITE_SIZE), instructions retired by kind (SQ_INSTS_*), LDS busy /
bank-conflict cycles, SQ_BUSY_CYCLES -- and the kernels' durations from the rocprofv3 kernel trace of the same round BY PHASE (tools/
kernel_phases.py: median / p10 of the calls inside the timed steps and of the single-kernel loop, the spin-up left out), stamped with the
hash of seqoia_amd/csrc so that bench.py reports `roofline.traffic`, `insts_per_64px` and `clock_ghz_pmc_pass` only for the kernel source the
counters were taken from.  `clock_ghz_pmc_pass` = SQ_BUSY_CYCLES / shader engines / the kernel's duration IN THE SAME COUNTER PASS (the dispatch
rows carry timestamps): the clock the chip held in that pass -- two steps, cold, serialised by the counter collection --, an estimate of, not the
same as, the clock of the timed run."""
import csv, hashlib, json, os, re, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "seqoia_amd", "csrc")
h = hashlib.sha256()
for f in sorted(os.listdir(src)):
    h.update(open(os.path.join(src, f), "rb").read())
kern, cur, n_se = {}, None, 32
for ln in open(sys.argv[1]):
    if ln.startswith("shader_engines"):
        n_se = int(ln.split()[1])
    elif ln.startswith("sqoa::"):
        cur = ln.strip()
        kern[cur] = {}
    elif cur and ln.strip():
        m = re.match(r"\s+(\S+)\s+([0-9.]+)", ln)
        if m:
            kern[cur][m.group(1)] = float(m.group(2))
dur = {}
if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
    for r in csv.DictReader(open(sys.argv[3])):
        dur[r["Name"].split("(")[0].replace("void ", "")] = {"avg_ns": float(r["AverageNs"]), "min_ns": float(r["MinNs"]), "calls": int(r["Calls"])}
phases = json.load(open(sys.argv[4])) if len(sys.argv) > 4 and os.path.exists(sys.argv[4]) else {}
out_images, size = 1024, 1024
pick = lambda d, pre: next(((k, v) for k, v in d.items() if k.startswith(pre)), (None, None))
out = {"note": f"rocprofv3 --pmc passes (one counter group per pass: HBM bytes, SQ_INSTS_*, busy / LDS cycles) of bench.py on the headline batch (1024 x 1024x1024 RGBA), per launch; "
               f"gfx950: FETCH_SIZE counts 1/2 of streamed read bytes (MI355X_MICROARCH.md, HBM) -> hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; "
               f"clock_ghz_pmc_pass = SQ_BUSY_CYCLES / {n_se} shader engines / the kernel's duration in the same counter pass; rocprof_in_step_ms / rocprof_alone_ms = median, p10 of the kernel's calls inside the timed steps / in the single-kernel loop of the kernel-trace pass (the spin-up left out). "
               f"Full table: profiles/{sys.argv[2]}_pmc_summary.txt, profiles/{sys.argv[2]}_kernel_stats.csv (tools/profile_round.sh {sys.argv[2]})",
       "csrc_sha256": h.hexdigest(), "images": out_images, "size": size, "kernels": {}}
groups = out_images * size * size / 64.0
for name, pres in (("encode", ("sqoa::encode_kernel<8, false>",)), ("decode", ("sqoa::bal::decode_bal_kernel<8, 16", "sqoa::decode_wave_kernel<8, 0"))):
    kname, k, pre = None, None, None
    for pre in pres:   # (the decoder the plan picked for the headline batch: the first of these that ran)
        kname, k = pick(kern, pre)
        if k and "FETCH_SIZE" in k and "WRITE_SIZE" in k:
            break
    if not k or "FETCH_SIZE" not in k or "WRITE_SIZE" not in k:
        continue
    row = {"kernel": kname, "FETCH_SIZE_KB": k["FETCH_SIZE"], "WRITE_SIZE_KB": k["WRITE_SIZE"]}
    if all(c in k for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS")):
        row["insts_per_64px"] = {c[9:].lower(): round(k[c] / groups, 2) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_BRANCH", "SQ_INSTS_LDS")}
    if "SQ_LDS_BANK_CONFLICT" in k and k.get("SQ_ACTIVE_INST_LDS"):
        row["lds_conflict_cycles_per_active_cycle"] = round(k["SQ_LDS_BANK_CONFLICT"] / k["SQ_ACTIVE_INST_LDS"], 3)
    _, d = pick(dur, pre)
    if d:
        row["rocprof_avg_ms"], row["rocprof_min_ms"], row["rocprof_calls"] = round(d["avg_ns"] / 1e6, 4), round(d["min_ns"] / 1e6, 4), d["calls"]
    if "SQ_BUSY_CYCLES" in k and "duration_ns@SQ_BUSY_CYCLES" in k:
        row["clock_ghz_pmc_pass"] = round(k["SQ_BUSY_CYCLES"] / n_se / k["duration_ns@SQ_BUSY_CYCLES"], 3)
    ph = phases.get({"encode": "encode"}.get(name, "decode_bal" if "bal" in (kname or "") else "decode_wave"))
    if ph:
        row["rocprof_in_step_ms"], row["rocprof_alone_ms"] = ph["in_step_ms"], ph["alone_ms"]
    out["kernels"][name] = row
json.dump(out, open(os.path.join(root, "profiles", "pmc_hbm_latest.json"), "w"), indent=1)
print(json.dumps(out["kernels"]))
