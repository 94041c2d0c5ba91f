#!/bin/bash
# tools/ab_d.sh "v1 v2 ...": SQOA decode A/B of variants/*.so: headline batch (two alternating rounds), icons RGB to 4 and 3 channels, photo RGB to 3
vs=$1
for r in 1 2; do for v in $vs; do
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py 2>/dev/null || exit 1
done; done
for v in $vs; do
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py --images 4096 --size 512 --channels 3 --cls icon --out-channels 4 2>/dev/null || exit 1
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py --images 4096 --size 512 --channels 3 --cls icon --out-channels 3 2>/dev/null || exit 1
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py --channels 3 --cls photo --out-channels 3 2>/dev/null || exit 1
done
