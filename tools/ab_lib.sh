#!/bin/bash
# tools/ab_lib.sh "v1 v2 ..." ROUNDS [time_decode.py | time_encode.py args, first word = decode|encode]: A/B of variants/*.so in alternating
# rounds on one box; every line printed, then per variant median / min / max of the ms per launch and the band (max - min) / median.
# A variant name may be a family: NAME@ stands for NAME_p0 NAME_p8 NAME_p16 (the same source at three code placements, tools/mkplaced.sh),
# reported both per placement and pooled -- a difference between two families counts only where it exceeds the placement band.
set -o pipefail
vs=$1; rounds=${2:-6}; what=${3:-decode}; shift 3
list=""
for v in $vs; do case $v in *@) b=${v%@}; list="$list ${b}_p0 ${b}_p8 ${b}_p16";; *) list="$list $v";; esac; done
tmp=$(mktemp)
for r in $(seq $rounds); do for v in $list; do
  out=$(SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_$what.py "$@" 2>/dev/null | tail -1) || exit 1
  echo "$out"; echo "$v $(echo "$out" | sed -n 's/.* \([0-9.]*\) ms.*/\1/p')" >> $tmp
done; done
python3 - $tmp <<'PY'
import sys, statistics as st, collections, re
d = collections.OrderedDict()
for l in open(sys.argv[1]):
    v, ms = l.split(); d.setdefault(v, []).append(float(ms))
fam = collections.OrderedDict()
for v, x in d.items():
    print(f"  {v}: median {st.median(x):.4f} ms  min {min(x):.4f}  max {max(x):.4f}  band {100 * (max(x) - min(x)) / st.median(x):.2f} %  (n={len(x)})")
    m = re.match(r"(.*)_p(0|8|16)$", v)
    if m: fam.setdefault(m.group(1), []).extend(x)
for f, x in fam.items():
    print(f"  {f}@ pooled over placements: median {st.median(x):.4f} ms  min {min(x):.4f}  max {max(x):.4f}  band {100 * (max(x) - min(x)) / st.median(x):.2f} %  (n={len(x)})")
PY
rm -f $tmp
