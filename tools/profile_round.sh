#!/bin/bash
# tools/profile_round.sh TAG   (on the GPU box, from the repo root)
# bench line + rocprofv3 kernel stats (20 warm-up + 20 timed steps, so that AverageNs is a warm average) + three
# separate PMC passes of the headline batch -> gpurun_out/TAG_*
set -e
tag=$1
out=gpurun_out
export TMPDIR=/tmp
python3 bench.py > $out/${tag}_bench.json 2> $out/${tag}_bench.err
rocprofv3 --output-format csv --kernel-trace --stats -d $out/prof_$tag -o run -- python3 bench.py --steps 20 --warmup 20 --no-cpu --no-secondary > $out/${tag}_prof.log 2>&1
rocprofv3 --output-format csv --pmc FETCH_SIZE -d $out/pmc_fetch_$tag -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
rocprofv3 --output-format csv --pmc WRITE_SIZE -d $out/pmc_write_$tag -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
rocprofv3 --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_BRANCH SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY -d $out/pmc_sq_$tag -o run -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-secondary > /dev/null 2>&1
python3 tools/pmc_table.py "$out/pmc_fetch_$tag/**/*counter_collection.csv" "$out/pmc_write_$tag/**/*counter_collection.csv" "$out/pmc_sq_$tag/**/*counter_collection.csv" > $out/${tag}_pmc_summary.txt
find $out/prof_$tag -name "*kernel_stats.csv" -exec cp {} $out/${tag}_kernel_stats.csv \;
cat $out/${tag}_bench.json; head -8 $out/${tag}_kernel_stats.csv
