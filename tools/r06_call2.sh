#!/bin/bash
# round 6, second session, call 2: SQOA_BAL_FLO_KEEP=3 with the rest in registers (fk3r) / the parse's look-ups through the vector cache (fk3g)
set -o pipefail
out=gpurun_out
mkdir -p $out
for v in fk3r fk3g; do
  SQOA_LIB=variants/$v.so timeout -k 10 300 python -m pytest tests/test_gpu_decode.py -x -q > $out/r06_k_tests_$v.txt 2>&1; echo "$v decode tests rc=$?"; tail -1 $out/r06_k_tests_$v.txt
done
timeout -k 10 600 tools/ab_lib.sh "base fk3 fk3r fk3g" 5 decode > $out/r06_k_ab_fk.txt 2>&1; tail -5 $out/r06_k_ab_fk.txt
