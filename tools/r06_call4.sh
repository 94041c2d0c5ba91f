#!/bin/bash
# round 6, second session, call 4: the new default (the parse's entries kept in registers, both decoders): the whole GPU suite, then A/B against the
# source of the round's first session on the headline batch, class photo, the icon set and the mixed-size suite's shard
set -o pipefail
out=gpurun_out
mkdir -p $out
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $out/r06_m_tests.txt 2>&1; echo "tests rc=$?"; tail -2 $out/r06_m_tests.txt
timeout -k 10 300 tools/ab_lib.sh "base new" 5 decode > $out/r06_m_ab_headline.txt 2>&1; tail -2 $out/r06_m_ab_headline.txt
timeout -k 10 300 tools/ab_lib.sh "base new" 4 decode --qoi 1 > $out/r06_m_ab_qoi.txt 2>&1; tail -2 $out/r06_m_ab_qoi.txt
timeout -k 10 300 tools/ab_lib.sh "base new" 3 decode --qoi 1 --cls photo > $out/r06_m_ab_qoi_photo.txt 2>&1; tail -2 $out/r06_m_ab_qoi_photo.txt
timeout -k 10 300 tools/ab_lib.sh "base new" 3 decode --images 4096 --size 512 --channels 3 --out-channels 3 --cls icon --qoi 1 > $out/r06_m_ab_qoi_icons.txt 2>&1; tail -2 $out/r06_m_ab_qoi_icons.txt
timeout -k 10 300 tools/ab_lib.sh "base new" 3 decode --channels 3 --out-channels 3 > $out/r06_m_ab_rgb.txt 2>&1; tail -2 $out/r06_m_ab_rgb.txt
for v in base new; do SQOA_LIB=variants/$v.so timeout -k 10 200 python tools/time_c5.py 2>/dev/null | tail -2; done > $out/r06_m_c5.txt; cat $out/r06_m_c5.txt
