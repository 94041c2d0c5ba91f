#!/bin/bash
# round 6, second session, call 1: the GPU suite on the rebuilt library, then the A/B of SQOA_BAL_FLO_KEEP and its parity
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $out/r06_j_tests.txt 2>&1; echo "tests rc=$?"; tail -2 $out/r06_j_tests.txt
for v in fk3 fk2; do
  SQOA_LIB=variants/$v.so timeout -k 10 300 python -m pytest tests/test_gpu_decode.py -x -q > $out/r06_j_tests_$v.txt 2>&1; echo "$v decode tests rc=$?"; tail -1 $out/r06_j_tests_$v.txt
done
timeout -k 10 500 tools/ab_lib.sh "base fk3 fk2" 5 decode > $out/r06_j_ab_fk.txt 2>&1; tail -4 $out/r06_j_ab_fk.txt
timeout -k 10 200 tools/ab_lib.sh "base fk3" 3 decode --cls photo > $out/r06_j_ab_fk_photo.txt 2>&1; tail -3 $out/r06_j_ab_fk_photo.txt
