#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 240 python -u tools/split_probe.py 8 > gpurun_out/r2f_split.log 2>&1 || { tail -30 gpurun_out/r2f_split.log; exit 1; }
cat gpurun_out/r2f_split.log
VX_LIB=$PWD/vista_amd/libvx_alt.so timeout -k 10 240 python -u tools/split_probe.py 8 > gpurun_out/r2f_split_alt.log 2>&1 || { tail -30 gpurun_out/r2f_split_alt.log; exit 1; }
cat gpurun_out/r2f_split_alt.log
