#!/bin/bash
# round 4, first A/B: wave shapes (W x VGPR budget) and stream bytes per lane of the SQOA decoder; wave shapes of the encoder
set -e
out=gpurun_out/r4a.txt
: > $out
for r in 1 2; do
  for v in base d7_8 d6_8 d6_12 d7_12 d5_16; do SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py >> $out 2>&1; done
  for v in base e7 e6; do SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_encode.py >> $out 2>&1; done
done
for v in base d6_12 d7_8; do SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py --cls icon --size 512 --images 4096 >> $out 2>&1; done
cat $out
SQOA_LIB=variants/d6_12.so timeout -k 10 900 python -m pytest tests/test_gpu_decode.py -x -q -m gpu > gpurun_out/r4a_test_d6_12.txt 2>&1; tail -5 gpurun_out/r4a_test_d6_12.txt
