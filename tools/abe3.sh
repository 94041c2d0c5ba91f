#!/bin/bash
# tools/abe3.sh "v1 v2 ...": encode-only A/B of variants/*.so on the three encode workloads (mixed RGBA, icons RGB, photo RGB)
for v in $1; do
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py || exit 1
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py --cls icon --channels 3 --images 4096 --size 512 || exit 1
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py --cls photo --channels 3 || exit 1
done
