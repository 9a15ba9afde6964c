#!/bin/bash
# round 6, call a: the fused linear-attention kernel in isolation (ablations + per-stage stamps, VERDICT r5 #1c) and a baseline bench of the round-5 tree on this box
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
./tools/mbench_la.bin > gpurun_out/r06_a_mbench_la.txt 2>&1
./tools/mbench_la.bin s >> gpurun_out/r06_a_mbench_la.txt 2>&1
cat gpurun_out/r06_a_mbench_la.txt
python3 bench.py --steps 2 --warmup 1 --T 100 --no-cpu-baseline > gpurun_out/r06_a_bench_T100.json 2> gpurun_out/r06_a_bench_T100.log
python3 -c "
import json; r=json.load(open('gpurun_out/r06_a_bench_T100.json')); print('T100 ms/step', r['ms_per_step']/100, 'launches', r['config'].get('launches_per_denoising_step'))"
