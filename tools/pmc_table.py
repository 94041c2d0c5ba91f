#!/usr/bin/env python3
"""Average rocprofv3 --pmc counter values per kernel from counter_collection.csv files."""
import csv, collections, glob, sys
d = collections.defaultdict(lambda: collections.defaultdict(list))
for pat in sys.argv[1:]:
    for f in glob.glob(pat, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            if "sqoa::" in k:
                d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in d.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:28s} {sum(v)/len(v):16.1f}  (n={len(v)})")
