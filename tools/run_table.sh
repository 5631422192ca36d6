#!/bin/bash
# The runs behind DESIGN.md section 5's table (through gpurun from the repo root: bash tools/run_table.sh <tag>)
set -e
t=$1
run() { local n=$1; shift; timeout -k 10 400 python3 bench.py "$@" > gpurun_out/${t}_$n.json 2> gpurun_out/${t}_$n.err; echo "$n done"; }
run default
run bf16 --dtype bf16 --no-cpu-baseline
run vista --geometry VISTA --backbone MultiViewBackbone --batch 4 --no-cpu-baseline
run full --geometry VISTA --backbone MultiViewBackbone --full-net --batch 4 --no-cpu-baseline --no-variants
run stress --geometry STRESS --batch 1 --no-cpu-baseline --no-variants
run train --mode train --batch 4
