#!/usr/bin/env python3
"""Static instruction counts of a kernel's hot loop BY REGION: compiles a .hip file with -DSQOA_ASM_MARK (the kernels' phase markers
become comments in the assembly), cuts the named kernel at the markers and prints, per instance of the loop body and per phase,
VALU (fast / slow class, tools/asm_cost.py's classes), scalar instructions, branches, LDS and global-memory instructions.
Static = every path of a region counted once, taken or not; the dynamic numbers per 64 pixels are in profiles/*pmc_summary.txt.
usage: asm_regions.py seqoia_amd/csrc/sqoa_decode_wave.hip '_ZN4sqoa18decode_wave_kernelILi8ELi0ELi12ELb0EEE' [names of the phases ...]"""
import os, re, subprocess, sys, tempfile
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asm_cost import classify
src, sym = sys.argv[1], sys.argv[2]
names = sys.argv[3:]
with tempfile.TemporaryDirectory() as d:
    out = os.path.join(d, "k.s")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-DSQOA_ASM_MARK",
                           "-S", "--cuda-device-only", "-o", out, src], stderr=subprocess.DEVNULL)
    lines = open(out).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith(sym) and l.rstrip().endswith(":") or l.startswith(sym) and ":" in l[:len(sym) + 200] and not l.startswith("\t"))
end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i])
marks = [(i, int(m.group(1))) for i in range(start, end) for m in [re.search(r"; SQOA_[A-Z]*PHASE (\d+)", lines[i])] if m]
print(f"{src}: {sym}\n{end - start} lines of assembly, {len(marks)} phase markers")
# (a marker closes its region, as in the timing builds: region k = the code between the marker in front and marker k)
hdr = f"{'instance':>8} {'region':<24} {'VALU':>5} {'fast':>5} {'slow':>5} {'SALU':>5} {'branch':>6} {'LDS':>4} {'VMEM':>5}"
print(hdr)
inst, tot = 0, {}
for k in range(1, len(marks)):
    (l0, p0), (l1, p1) = marks[k - 1], marks[k]
    cnt = {}
    for l in lines[l0:l1]:
        c = classify(l)
        if c: cnt[c] = cnt.get(c, 0) + 1
    fast, slow = cnt.get("fast", 0), cnt.get("slow", 0) + cnt.get("slow_sgpr", 0)
    name = names[p1] if p1 < len(names) else f"phase {p1}"
    row = (fast + slow, fast, slow, cnt.get("salu", 0), cnt.get("branch", 0), cnt.get("lds", 0), cnt.get("vmem", 0))
    print(f"{inst:>8} {name:<24} " + " ".join(f"{v:>5}" for v in row))
    tot[inst] = tuple(a + b for a, b in zip(tot.get(inst, (0,) * 7), row))
    if p1 < p0 and k + 1 < len(marks):   # the markers wrap: the next instance of the loop body
        print(f"{inst:>8} {'= this instance':<24} " + " ".join(f"{v:>5}" for v in tot[inst]))
        inst += 1
print(f"{inst:>8} {'= this instance':<24} " + " ".join(f"{v:>5}" for v in tot.get(inst, (0,) * 7)))
