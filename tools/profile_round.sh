#!/bin/bash
# tools/profile_round.sh TAG   (on the GPU box, from the repo root)
# bench line + rocprofv3 kernel stats (20 warm-up + 20 timed steps, so that AverageNs is a warm average) + separate PMC
# passes (HBM bytes, SQ issue counters, LDS busy / bank-conflict cycles) of the headline batch; then the same kernel stats
# + SQ + LDS passes for the QOI-compat kernels on the headline batch and for the RGB icon batch (BASELINE configs[3]) in
# both wire formats, and the kernel stats of one large QOI-compat stream alone -> gpurun_out/TAG_*   (copy what is to be judged into profiles/)
set -e
tag=$1
out=gpurun_out
export TMPDIR=/tmp
SQ1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY"
SQ2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT"
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof_$tag -o run -- python3 bench.py --steps 20 --warmup 20 --no-cpu --no-secondary > $out/${tag}_prof.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/pmc_fetch_$tag -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/pmc_write_$tag -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
rocprofv3 --output-format csv --pmc $SQ1 -d $out/pmc_sq_$tag -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
rocprofv3 --output-format csv --pmc $SQ2 -d $out/pmc_sq2_$tag -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
python3 tools/pmc_table.py "$out/pmc_fetch_$tag/**/*counter_collection.csv" "$out/pmc_write_$tag/**/*counter_collection.csv" "$out/pmc_sq_$tag/**/*counter_collection.csv" "$out/pmc_sq2_$tag/**/*counter_collection.csv" > $out/${tag}_pmc_summary.txt
find $out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
# the kernels' durations by phase of that run (median / p10 of the 20 calls inside the timed steps and of the 20 of the single-kernel loop: the
# mean of the stats file runs over the spin-up as well); back in the build container: tools/pmc_latest.py gpurun_out/TAG_pmc_summary.txt TAG
# gpurun_out/TAG_kernel_stats.csv gpurun_out/TAG_kernel_phases.json -> profiles/pmc_hbm_latest.json
python3 tools/kernel_phases.py $(find $out/prof_$tag -name "*kernel_trace.csv" | head -1) 20 > $out/${tag}_kernel_phases.json
# workgroup timelines of the two headline kernels (start / end of every workgroup, per compute unit: median / p99 / max end time) -- from the
# diagnostic build, when the snapshot carries one (tools/mkvariant.sh trace "-DSQOA_TRACE" in the build container; its times are the
# instrumented build's, ~40 % / 15 % longer than the product's: read the SHAPE of the distribution)
if [ -f variants/trace.so ]; then
    SQOA_LIB=variants/trace.so python3 tools/trace_encode.py > $out/${tag}_trace_encode.txt 2>/dev/null || true
    SQOA_LIB=variants/trace.so python3 tools/trace_decode.py > $out/${tag}_trace_decode.txt 2>/dev/null || true
fi
# secondary workloads: tools/prof_kernels.py launches only the product kernels
sec() {   # sec NAME prof_kernels-args...
    name=$1; shift
    rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof_${tag}_$name -o run -- python3 tools/prof_kernels.py --iters 10 "$@" > $out/${tag}_${name}_prof.log 2>&1
    rocprofv3 --output-format csv --pmc $SQ1 -d $out/pmc_sq_${tag}_$name -o run -- python3 tools/prof_kernels.py --iters 2 "$@" > /dev/null 2>&1
    rocprofv3 --output-format csv --pmc $SQ2 -d $out/pmc_sq2_${tag}_$name -o run -- python3 tools/prof_kernels.py --iters 2 "$@" > /dev/null 2>&1
    find $out/prof_${tag}_$name -name "*kernel_stats.csv" -exec cp {} $out/${tag}_${name}_kernel_stats.csv \;
    python3 tools/pmc_table.py "$out/pmc_sq_${tag}_$name/**/*counter_collection.csv" "$out/pmc_sq2_${tag}_$name/**/*counter_collection.csv" > $out/${tag}_${name}_pmc_summary.txt
}
sec qoi --images 1024 --qoi 1
sec icons --images 4096 --size 512 --channels 3 --cls icon --qoi 0
sec icons_qoi --images 4096 --size 512 --channels 3 --cls icon --qoi 1
# one large QOI-compat stream alone (2880 x 2880 = the pixels of 3840 x 2160): the chip-wide routes -- strips for the encode,
# passes 1 and 2 + the speculative pass + its check for the decode; kernel stats only
rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof_${tag}_lone_qoi -o run -- python3 tools/prof_kernels.py --iters 10 --images 1 --size 2880 --qoi 1 > $out/${tag}_lone_qoi_prof.log 2>&1
find $out/prof_${tag}_lone_qoi -name "*kernel_stats.csv" -exec cp {} $out/${tag}_lone_qoi_kernel_stats.csv \;
cat $out/${tag}_bench.json; head -8 $out/${tag}_kernel_stats.csv
