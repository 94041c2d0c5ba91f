#!/bin/bash
# tools/ab_env.sh VAR "v1 v2 ..." ROUNDS [time_decode.py args...]: A/B of ONE library under different values of an environment knob
# (e.g. SQOA_HIP_DEC_BAL "0 1" 6): alternating rounds, every line printed, then per value median / min / max of the ms per launch.
var=$1; vals=$2; rounds=${3:-6}; shift 3
tmp=$(mktemp)
for r in $(seq $rounds); do for v in $vals; do
  out=$(env $var=$v timeout -k 10 120 python tools/time_decode.py "$@" 2>/dev/null) || exit 1
  echo "$var=$v $out"; echo "$v $(echo "$out" | sed -n 's/.*decode \([0-9.]*\) ms.*/\1/p')" >> $tmp
done; done
python3 - $tmp <<'PY'
import sys, statistics as st, collections
d = collections.defaultdict(list)
for l in open(sys.argv[1]):
    v, ms = l.split(); d[v].append(float(ms))
for v, x in d.items(): print(f"  {v}: median {st.median(x):.4f} ms  min {min(x):.4f}  max {max(x):.4f}  (n={len(x)})")
PY
rm -f $tmp
