#!/bin/bash
set -e
out=gpurun_out/r4d.txt
: > $out
for v in base n12; do
  echo "== $v" >> $out
  SQOA_LIB=variants/$v.so timeout -k 10 300 python tools/perf_table.py 2>/dev/null >> $out
  SQOA_LIB=variants/$v.so timeout -k 10 300 python tools/time_c5.py 2>/dev/null >> $out
  SQOA_LIB=variants/$v.so timeout -k 10 300 python tools/time_rgb.py 2>/dev/null >> $out
done
cat $out
