#!/bin/bash
# tools/pmc_sq.sh TAG [tool args]  (on the GPU box): SQ issue counters of the encode + decode kernels on the headline batch
set -e
tag=$1; shift
out=gpurun_out
export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/pmc_sq_$tag -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
python3 tools/pmc_table.py "$out/pmc_sq_$tag/**/*counter_collection.csv" > $out/${tag}_sq.txt
cat $out/${tag}_sq.txt
