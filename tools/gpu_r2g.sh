#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
VX_PROBE_TIME_ONLY=1 timeout -k 10 240 python -u tools/split_probe.py 8 > gpurun_out/r2g_split.log 2>&1 || { tail -30 gpurun_out/r2g_split.log; exit 1; }
grep -E "products" gpurun_out/r2g_split.log | cut -c1-60
for lib in libvx_abl9; do
  echo "== $lib"
  VX_PROBE_TIME_ONLY=1 VX_LIB=$PWD/vista_amd/$lib.so timeout -k 10 120 python -u tools/split_probe.py 8 > gpurun_out/r2g_$lib.log 2>&1 || { tail -30 gpurun_out/r2g_$lib.log; exit 1; }
  grep -E "products" gpurun_out/r2g_$lib.log | cut -c1-60
done
