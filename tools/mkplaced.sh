#!/bin/bash
# tools/mkplaced.sh NAME "-DFLAG=.. ..."  -> variants/NAME_p0.so, NAME_p8.so, NAME_p16.so: the same source and flags with the kernels' bodies
# 0 / 8 / 16 bytes further down (-DSQOA_PAD = 0 / 2 / 4 s_nop at the kernels' entry, sqoa_common.h).  tools/ab_lib.sh "A@ B@" runs the two
# families in alternating rounds and reports every placement and the pooled band.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
for p in 0 2 4; do tools/mkvariant.sh ${name}_p$((p * 4)) "$@ -DSQOA_PAD=$p" | tail -1; done
