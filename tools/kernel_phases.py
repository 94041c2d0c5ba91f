#!/usr/bin/env python3
"""tools/kernel_phases.py KERNEL_TRACE.csv [steps] -> the headline kernels' durations BY PHASE of a `bench.py --steps K --warmup W` run under
rocprofv3 --kernel-trace: bench.py runs untimed spin-up steps (the clocks ramp), W warm-up steps, K timed steps (encode, decode, encode, ...)
and then, for the per-kernel `plan_launch_ms`, K launches of the encoder alone and K of the decoder alone.  A mean over all of a kernel's
calls mixes the ramp into a steady-state time; here: median / p10 / p90 of the K calls inside the timed steps (`in_step`) and of the K calls
of the single-kernel loop (`alone`), per kernel, as JSON on stdout (and what the spin-up's first and last eight calls took)."""
import csv, json, sys
import statistics as st
K = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows = [r for r in csv.DictReader(open(sys.argv[1])) if r.get("Kind", "KERNEL_DISPATCH") == "KERNEL_DISPATCH"]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
heads = {"encode": "sqoa::encode_kernel<8, false>", "decode_bal": "sqoa::bal::decode_bal_kernel<8, 16", "decode_wave": "sqoa::decode_wave_kernel<8, 0"}
out = {}
def q(x, p):
    x = sorted(x); return x[min(len(x) - 1, int(p * len(x)))]
for name, pre in heads.items():
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows if r["Kernel_Name"].replace("void ", "").startswith(pre)]
    if len(d) < 2 * K + 8:
        continue
    alone, step, spin = d[-K:], d[-2 * K:-K], d[:-2 * K]
    f = lambda x: {"median": round(st.median(x), 4), "p10": round(q(x, 0.1), 4), "p90": round(q(x, 0.9), 4), "n": len(x)}
    out[name] = {"in_step_ms": f(step), "alone_ms": f(alone), "spin_up_first8_ms": round(st.mean(spin[:8]), 4), "spin_up_last8_ms": round(st.mean(spin[-8:]), 4), "calls": len(d)}
print(json.dumps(out, indent=1))
