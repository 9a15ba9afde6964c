#!/bin/bash
# usage: tools/gpu_lib_ab2.sh <tag> <other-lib.so> [reps]  -- same-box interleaved A/B of the in-tree libddif.so against another build of it (bench.py --lib)
tag=$1; other=$2; reps=${3:-3}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for rep in $(seq 1 $reps); do
  for which in other tree; do
    if [ $which = other ]; then L="--lib $other"; else L=""; fi
    python3 bench.py $L --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('$which rep $rep ms/step', round(r['ms_per_step']/200,4), 'launches', r['config']['launches_per_denoising_step'])" | tee -a gpurun_out/${tag}_lib_ab.txt
  done
done
