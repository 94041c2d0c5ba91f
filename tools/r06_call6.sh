#!/bin/bash
# round 6, second session, call 6: the GPU suite and the profile round of the final source in one call
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $out/r06_p_tests.txt 2>&1; echo "tests rc=$?"; tail -2 $out/r06_p_tests.txt
timeout -k 10 900 tools/profile_round.sh r06_p > $out/r06_p_round.log 2>&1; echo "profile rc=$?"; tail -12 $out/r06_p_round.log
