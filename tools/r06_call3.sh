#!/bin/bash
# round 6, second session, call 3: how many of the sixteen entries in registers (fk{3,2,1,0}r); the QOI-compat decoder's seven kept (q1: registers, q2: packed)
set -o pipefail
out=gpurun_out
mkdir -p $out
for v in q1 q2 fk0r fk1r; do
  SQOA_LIB=variants/$v.so timeout -k 10 300 python -m pytest tests/test_gpu_decode.py -x -q > $out/r06_l_tests_$v.txt 2>&1; echo "$v decode tests rc=$?"; tail -1 $out/r06_l_tests_$v.txt
done
timeout -k 10 600 tools/ab_lib.sh "base fk3r fk2r fk1r fk0r" 5 decode > $out/r06_l_ab_fkr.txt 2>&1; tail -6 $out/r06_l_ab_fkr.txt
timeout -k 10 600 tools/ab_lib.sh "base q1 q2" 5 decode --qoi 1 > $out/r06_l_ab_q.txt 2>&1; tail -4 $out/r06_l_ab_q.txt
