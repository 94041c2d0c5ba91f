#!/usr/bin/env python3
"""Static issue-cost estimate of gfx950 assembly (from hipcc -S): VALU instructions weighted by the two cost classes
measured with tools/ubench/valu_table.hip -- 'fast' (add/sub/and/or/xor/not/mov/lshrrev/ashrrev/bitop3 whose operands
are VGPRs, inline constants or literals) = 1.0, every other VALU (and any of the above with an SGPR operand) = 1.73.
usage: asm_cost.py file.s [first_line last_line]"""
import re, sys
FAST = {"v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32", "v_not_b32", "v_mov_b32",
        "v_lshrrev_b32", "v_ashrrev_i32", "v_bitop3_b32"}
def classify(line):
    t = line.split(";")[0].strip()
    if not t or t.endswith(":") or t.startswith("."): return None
    op = t.split()[0]
    if op.startswith("v_"):
        base = re.sub(r"_(e32|e64)$", "", op)
        if "sdwa" in op or "dpp" in op or base not in FAST: return "slow"
        ops = t[len(op):]
        ops = re.sub(r"bitop3:\S+", "", ops)
        srcs = ops.split(",")[1:]
        if any(re.search(r"\b(s\d+|s\[\d+:\d+\]|vcc|exec|m0|ttmp\d+)", x) for x in srcs): return "slow_sgpr"
        return "fast"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch"): return "branch"
    if op.startswith("s_"): return "salu"
    if op.startswith("ds_"): return "lds"
    if op.startswith("global_") or op.startswith("scratch_") or op.startswith("buffer_") or op.startswith("flat_"): return "vmem"
    return "other"
def main():
    lines = open(sys.argv[1]).read().split("\n")
    a, b = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1, len(lines))
    cnt, slow_ops = {}, {}
    for ln in lines[a - 1:b]:
        c = classify(ln)
        if c is None: continue
        cnt[c] = cnt.get(c, 0) + 1
        if c in ("slow", "slow_sgpr"):
            k = ln.split()[0] + (" [sgpr]" if c == "slow_sgpr" else "")
            slow_ops[k] = slow_ops.get(k, 0) + 1
    valu = cnt.get("fast", 0) + cnt.get("slow", 0) + cnt.get("slow_sgpr", 0)
    cost = cnt.get("fast", 0) + 1.73 * (cnt.get("slow", 0) + cnt.get("slow_sgpr", 0))
    print(f"lines {a}-{b}: VALU {valu} (fast {cnt.get('fast',0)}, slow {cnt.get('slow',0)}, fast-op-with-SGPR {cnt.get('slow_sgpr',0)}) cost {cost:.0f}; "
          f"SALU {cnt.get('salu',0)} branch {cnt.get('branch',0)} LDS {cnt.get('lds',0)} VMEM {cnt.get('vmem',0)} wait/nop {cnt.get('wait',0)}")
    for k, v in sorted(slow_ops.items(), key=lambda x: -x[1])[:25]: print(f"   {v:5d} {k}")
if __name__ == "__main__":
    main()
