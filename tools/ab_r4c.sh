#!/bin/bash
set -e
out=gpurun_out/r4c.txt
: > $out
for r in 1 2; do
  for v in base n8 n12; do SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py 2>/dev/null >> $out; done
done
for v in base n8 n12; do SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py --cls icon --size 512 --images 4096 2>/dev/null >> $out; done
for v in base n8 n12; do SQOA_LIB=variants/$v.so timeout -k 10 120 python tools/time_decode.py --cls photo 2>/dev/null >> $out; done
cat $out
SQOA_LIB=variants/n12.so timeout -k 10 900 python -m pytest tests/test_gpu_decode.py tests/test_gpu_encode.py -x -q -m gpu > gpurun_out/r4c_test_n12.txt 2>&1; tail -5 gpurun_out/r4c_test_n12.txt
