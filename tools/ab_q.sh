#!/bin/bash
# tools/ab_q.sh "v1 v2 ...": QOI-compat decode A/B of variants/*.so, headline batch and icons, two alternating rounds
vs=$1
for r in 1 2; do for v in $vs; do
  SQOA_LIB=variants/$v.so timeout -k 10 200 python tools/time_qoi_decode.py 2>/dev/null || exit 1
done; done
for v in $vs; do SQOA_LIB=variants/$v.so timeout -k 10 200 python tools/time_qoi_icons.py 2>/dev/null || exit 1; done
