#!/bin/bash
set -o pipefail
mkdir -p gpurun_out
for lib in libvx libvx_abl1 libvx_abl2 libvx_abl3 libvx_abl4 libvx_abl5; do
  VX_LIB=$PWD/vista_amd/$lib.so timeout -k 10 200 python -u tools/conv_probe.py 8 nocheck > gpurun_out/r2n_$lib.log 2>&1 || { tail gpurun_out/r2n_$lib.log; exit 1; }
  echo "== $lib"; grep -E "subm2|conv3" gpurun_out/r2n_$lib.log | cut -c1-110
done
