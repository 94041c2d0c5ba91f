#!/bin/bash
# tools/pmc_sq2.sh TAG (on the GPU box): busy-cycle counters of the headline kernels
set -e
tag=$1; out=gpurun_out; export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT -d $out/pmc_sq2_$tag -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
python3 tools/pmc_table.py "$out/pmc_sq2_$tag/**/*counter_collection.csv" > $out/${tag}_sq2.txt
cat $out/${tag}_sq2.txt
