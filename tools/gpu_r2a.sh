set -o pipefail
mkdir -p gpurun_out/r2a
timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -x -v --timeout 120 --timeout-method thread > gpurun_out/r2a/noalloc.log 2>&1 ; echo "noalloc rc=$?" 
timeout -k 10 400 python -u -m pytest tests/test_gpu_parity.py -x -q --timeout 120 --timeout-method thread -k "backbone_forward" > gpurun_out/r2a/parity.log 2>&1 ; echo "parity rc=$?"
timeout -k 10 300 python -u bench.py --steps 5 --warmup 2 --no-variants --no-cpu-baseline > gpurun_out/r2a/bench1.log 2>&1 ; echo "bench1 rc=$?"
VX_DIST_BACKEND=nccl timeout -k 10 200 python -u bench.py --gpus 2 --steps 3 --warmup 1 --no-variants --no-cpu-baseline --batch 2 > gpurun_out/r2a/bench2_nccl.log 2>&1 ; echo "bench2 nccl rc=$?"
timeout -k 10 300 python -u bench.py --gpus 2 --steps 3 --warmup 1 --no-variants --no-cpu-baseline --batch 2 > gpurun_out/r2a/bench2.log 2>&1 ; echo "bench2 rc=$?"
timeout -k 10 300 python -u bench.py --gpus 2 --mode train --steps 2 --warmup 1 --batch 1 > gpurun_out/r2a/train2.log 2>&1 ; echo "train2 rc=$?"
tail -3 gpurun_out/r2a/*.log
