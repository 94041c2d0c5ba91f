#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel from counter_collection.csv files.  Beside every counter group's values the
kernel's mean duration IN THAT PASS (End - Start of the same dispatch rows, `duration_ns@COUNTER`): cycles from one pass over durations
from another is not a clock -- an HBM-bound kernel's cycles are not clock-invariant, and a counter pass runs cold and serialised.
And, from the agent_info.csv next to a file, the device's shader-engine count (`shader_engines`: what SQ_BUSY_CYCLES is summed over)."""
import csv, collections, glob, os, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
dur = collections.defaultdict(lambda: collections.defaultdict(dict))   # kernel -> counter -> {dispatch id: ns}
ses = None
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "sqoa::" in k:
                d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                if r.get("Start_Timestamp") and r.get("End_Timestamp"):
                    dur[k][r["Counter_Name"]][(f, r["Dispatch_Id"])] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        a = os.path.join(os.path.dirname(f), os.path.basename(f).replace("counter_collection", "agent_info"))
        if ses is None and os.path.exists(a):
            for r in csv.DictReader(open(a)):
                if r.get("Agent_Type") == "GPU":
                    ses = int(r.get("Array_Count") or 0) or None
                    break
if ses:
    print(f"shader_engines {ses}")
for k, cs in d.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
    for c in ("SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "FETCH_SIZE", "WRITE_SIZE"):
        if c in dur[k]:
            x = list(dur[k][c].values())
            print(f"   {'duration_ns@' + c:28s} {sum(x)/len(x):16.1f}  (n={len(x)})")
