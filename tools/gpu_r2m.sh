#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 240 python -u tools/split_probe.py 8 > gpurun_out/r2m_split.log 2>&1 || { tail -30 gpurun_out/r2m_split.log; exit 1; }
grep -E "products" gpurun_out/r2m_split.log | cut -c1-140
VX_PROBE_TIME_ONLY=1 VX_LIB=$PWD/vista_amd/libvx_alt.so timeout -k 10 240 python -u tools/split_probe.py 8 > gpurun_out/r2m_split_alt.log 2>&1 || { tail -30 gpurun_out/r2m_split_alt.log; exit 1; }
echo "== 32x32x16"; grep -E "products" gpurun_out/r2m_split_alt.log | cut -c1-60
