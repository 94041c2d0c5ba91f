#!/bin/bash
set -e
out=gpurun_out/r4b.txt
: > $out
for r in 1 2; do
  for v in base d6w7_8 d6w7_12; do SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py 2>/dev/null >> $out; done
  for v in base e6w7; do SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py 2>/dev/null >> $out; done
done
cat $out
