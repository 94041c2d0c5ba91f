#!/bin/bash
# tools/timeline.sh TAG prof_kernels-args...  (on the GPU box): the dispatches of a plan's launches, one after the other, with the gaps between them
set -e
tag=$1; shift
out=gpurun_out; export TMPDIR=/tmp
rocprofv3 --output-format csv --kernel-trace -d $out/tl_$tag -o run -- python3 tools/prof_kernels.py --iters 10 "$@" > $out/tl_${tag}.log 2>&1
f=$(find $out/tl_$tag -name "*kernel_trace.csv" | head -1)
python3 tools/kernel_timeline.py $f 16 > $out/${tag}_timeline.txt
cat $out/${tag}_timeline.txt
