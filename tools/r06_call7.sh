#!/bin/bash
# round 6, second session, call 7: the balanced decoder's older knobs once more behind the register change (new = the tree)
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 700 tools/ab_lib.sh "new hl1 pw2 pl1 ss2 ss8" 4 decode > $out/r06_q_ab_knobs.txt 2>&1; tail -7 $out/r06_q_ab_knobs.txt
