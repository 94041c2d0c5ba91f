#!/bin/bash
# tools/mkvariant.sh NAME "-DFLAG=.. ..."  -> variants/NAME.so (A/B builds selected with SQOA_LIB=variants/NAME.so)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p variants build/v_$name
FILES="sqoa_encode sqoa_decode sqoa_decode_wave sqoa_decode_bal sqoa_decode_qoi sqoa_strips sqoa_runtime sqoa_selftest"
for f in $FILES; do
  rm -f build/v_$name/$f.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $@ -c seqoia_amd/csrc/$f.hip -o build/v_$name/$f.o &
done
# (the encoder once more: its QOI-compat kernels in the order-free form, Makefile)
rm -f build/v_$name/sqoa_encode_fb.o; /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $@ -DSQOA_LDS_ORDER_FB -c seqoia_amd/csrc/sqoa_encode.hip -o build/v_$name/sqoa_encode_fb.o &
wait
for f in $FILES sqoa_encode_fb; do [ -f build/v_$name/$f.o ] || { echo "mkvariant: $f.hip did not compile" >&2; exit 1; }; done
gcc -std=gnu99 -O2 -fPIC -c seqoia_amd/csrc/sqoa_host.c -o build/v_$name/sqoa_host.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o variants/$name.so build/v_$name/*.o -lpthread
echo built variants/$name.so
