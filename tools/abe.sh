#!/bin/bash
# tools/abe.sh "v1 v2 ..." [args]: encode-only same-call A/B of variants/*.so, two alternating rounds
vs=$1; shift
for r in 1 2; do for v in $vs; do SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py "$@" || exit 1; done; done
