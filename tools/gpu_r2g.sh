#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 240 python -u tools/split_probe.py 8 > gpurun_out/r2g_split.log 2>&1 || { tail -30 gpurun_out/r2g_split.log; exit 1; }
grep -E "products" gpurun_out/r2g_split.log
: > gpurun_out/r2g_abl.log
for lib in ${VX_ABL_LIBS:-libvx_abl3}; do
  echo "== $lib" >> gpurun_out/r2g_abl.log
  VX_PROBE_TIME_ONLY=1 VX_LIB=$PWD/vista_amd/$lib.so timeout -k 10 120 python -u tools/split_probe.py 8 >> gpurun_out/r2g_abl.log 2>&1 || { tail -30 gpurun_out/r2g_abl.log; exit 1; }
done
grep -E "==|products" gpurun_out/r2g_abl.log | cut -c1-60
