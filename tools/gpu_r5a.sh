#!/bin/bash
# round 5, first call: resident-weights conv micro-benchmark (mbench r) + kernarg placement A/B on the default bench (T = 200)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 300 $R/tools/mbench.bin r > $R/gpurun_out/r05_a_mbench_resident.txt 2>&1
cat $R/gpurun_out/r05_a_mbench_resident.txt | grep -v "^  block\|^    [0-9 -]" | head -60
for rep in 1 2; do
  for v in 0 1; do
    HIP_FORCE_DEV_KERNARG=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/r05_a_kernarg_${v}_$rep.json 2> /dev/null
    python3 -c "
import json; r=json.load(open('gpurun_out/r05_a_kernarg_${v}_$rep.json')); print('HIP_FORCE_DEV_KERNARG=$v', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'])"
  done
done
