#!/bin/bash
# round 6, second session, call 9: the QOI-compat decoder's poll intervals and the resolver's in-register threshold once more behind the producers' change
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 700 tools/ab_lib.sh "new qs8 qs32 qseq6 qrs2" 4 decode --qoi 1 > $out/r06_r_ab_qknobs.txt 2>&1; tail -6 $out/r06_r_ab_qknobs.txt
