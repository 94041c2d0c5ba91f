#!/bin/bash
# tools/ab.sh "v1 v2 ..." [extra args for time_*.py]   (same-call A/B of variants/*.so, two alternating rounds)
vs=$1; shift
for r in 1 2; do for v in $vs; do
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py "$@" || exit 1
  SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py "$@" || exit 1
done; done
