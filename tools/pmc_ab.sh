#!/bin/bash
# tools/pmc_ab.sh TAG "v1 v2 ..." prof_kernels-args...   (on the GPU box): SQ counters of the product kernels for variants/*.so, same workload
set -e
tag=$1; vs=$2; shift 2
out=gpurun_out; export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
SQ3="SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_IFETCH SQ_WAIT_INST_ANY SQ_INSTS_SMEM SQ_INSTS_VALU"
for v in $vs; do
  export SQOA_LIB=variants/$v.so
  rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof_${tag}_$v -o run -- python3 tools/prof_kernels.py --iters 10 "$@" > $out/${tag}_${v}_prof.log 2>&1
  rocprofv3 --output-format csv --pmc $SQ1 -d $out/pmc1_${tag}_$v -o run -- python3 tools/prof_kernels.py --iters 2 "$@" > /dev/null 2>&1
  rocprofv3 --output-format csv --pmc $SQ2 -d $out/pmc2_${tag}_$v -o run -- python3 tools/prof_kernels.py --iters 2 "$@" > /dev/null 2>&1
  rocprofv3 --output-format csv --pmc $SQ3 -d $out/pmc3_${tag}_$v -o run -- python3 tools/prof_kernels.py --iters 2 "$@" > /dev/null 2>&1 || true
  python3 tools/pmc_table.py "$out/pmc1_${tag}_$v/**/*counter_collection.csv" "$out/pmc2_${tag}_$v/**/*counter_collection.csv" "$out/pmc3_${tag}_$v/**/*counter_collection.csv" > $out/${tag}_${v}_pmc.txt
  find $out/prof_${tag}_$v -name "*kernel_stats.csv" -exec cp {} $out/${tag}_${v}_kernel_stats.csv \;
  echo "== $v"; (grep -A30 "decode_wave_kernel" $out/${tag}_${v}_pmc.txt || true) | head -32
done
