#!/bin/bash
# tools/ab_e.sh "v1 v2 ...": SQOA encode A/B of variants/*.so on the headline batch, icons RGB and photo RGB; two alternating rounds
vs=$1
for r in 1 2; do for v in $vs; do
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py 2>/dev/null || exit 1
done; done
for v in $vs; do
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py --images 4096 --size 512 --channels 3 --cls icon 2>/dev/null || exit 1
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py --channels 3 --cls photo 2>/dev/null || exit 1
done
