#!/bin/bash
# tools/fetch_size.sh LIB : FETCH_SIZE / WRITE_SIZE of the encode kernel (headline batch) with the given library
export TMPDIR=/tmp
rm -rf gpurun_out/fs_$$
SQOA_LIB=$1 rocprofv3 --output-format csv --pmc FETCH_SIZE -d gpurun_out/fs_$$ -o run -- python3 tools/prof_kernels.py --images 1024 --what encode > /dev/null 2>&1
python3 tools/pmc_table.py "gpurun_out/fs_$$/**/*counter_collection.csv" | grep -A1 "encode_kernel" | tr '\n' ' '; echo " <- $1"
rm -rf gpurun_out/fs_$$
