#!/bin/bash
# round 6, second session, call 8: workgroup / phase traces of the headline kernels on the final source (diagnostic build)
set -o pipefail
out=gpurun_out
mkdir -p $out
SQOA_LIB=variants/trace.so timeout -k 10 300 python3 tools/trace_decode.py > $out/r06_p_trace_decode.txt 2>$out/r06_p_trace_decode.err; echo "trace decode rc=$?"; cat $out/r06_p_trace_decode.txt
SQOA_LIB=variants/trace.so timeout -k 10 300 python3 tools/trace_encode.py > $out/r06_p_trace_encode.txt 2>$out/r06_p_trace_encode.err; echo "trace encode rc=$?"; tail -3 $out/r06_p_trace_encode.txt
