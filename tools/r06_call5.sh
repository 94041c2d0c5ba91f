#!/bin/bash
# round 6, second session, call 5: a lane's bytes for the parse from the registers they were fetched into (qdr: balanced SQOA decoder; qown: QOI-compat decoder)
set -o pipefail
out=gpurun_out
mkdir -p $out
for v in qdr qown; do
  SQOA_LIB=variants/$v.so timeout -k 10 300 python -m pytest tests/test_gpu_decode.py -x -q > $out/r06_n_tests_$v.txt 2>&1; echo "$v decode tests rc=$?"; tail -1 $out/r06_n_tests_$v.txt
done
timeout -k 10 300 tools/ab_lib.sh "new qdr" 5 decode > $out/r06_n_ab_qdr.txt 2>&1; tail -2 $out/r06_n_ab_qdr.txt
timeout -k 10 300 tools/ab_lib.sh "new qown" 4 decode --qoi 1 > $out/r06_n_ab_qown.txt 2>&1; tail -2 $out/r06_n_ab_qown.txt
timeout -k 10 300 tools/ab_lib.sh "new qown" 3 decode --qoi 1 --cls photo > $out/r06_n_ab_qown_photo.txt 2>&1; tail -2 $out/r06_n_ab_qown_photo.txt
timeout -k 10 300 tools/ab_lib.sh "new qown" 3 decode --images 4096 --size 512 --channels 3 --out-channels 3 --cls icon --qoi 1 > $out/r06_n_ab_qown_icons.txt 2>&1; tail -2 $out/r06_n_ab_qown_icons.txt
