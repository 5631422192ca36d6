#!/bin/bash
# two ranks on the one GPU of the box (gloo fallback): launcher form and torch.distributed.run form
set -o pipefail
mkdir -p gpurun_out
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 > gpurun_out/n2_infer.json 2> gpurun_out/n2_infer.err || { tail -20 gpurun_out/n2_infer.err; exit 1; }
cut -c1-330 gpurun_out/n2_infer.json
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --mode train > gpurun_out/n2_train.json 2> gpurun_out/n2_train.err || { tail -20 gpurun_out/n2_train.err; exit 1; }
tail -1 gpurun_out/n2_train.json | cut -c1-330
