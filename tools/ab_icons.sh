#!/bin/bash
# tools/ab_icons.sh "v1 v2 ...": RUN-heavy decode A/B -- icons 4096 x 512 x 512 RGB to 3 and 4 channels (tools/time_rgb.py), SQOA and QOI-compat
for v in $1; do echo "== $v"; SQOA_LIB=variants/$v.so timeout -k 10 200 python tools/time_rgb.py 2>/dev/null | grep -v encode; SQOA_LIB=variants/$v.so timeout -k 10 200 python tools/time_qoi_icons.py 2>/dev/null; done
