#!/bin/bash
# round 6, call g: linattn_fused with the transposed-q softmax section (in-register column reductions, one pair exchange) -- microbenchmark, parity, bench
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(./tools/mbench_la.bin | grep -E "abl= 0|abl=16"; ./tools/mbench_la.bin s | tail -6) > gpurun_out/r06_g_mbench_la.txt 2>&1; cut -c1-400 gpurun_out/r06_g_mbench_la.txt
(python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r06_g_tests.log
cat gpurun_out/r06_g_tests.log
for rep in 1 2; do
  python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('rep $rep ms/step', round(r['ms_per_step']/200,4), 'launches', r['config']['launches_per_denoising_step'])" | tee -a gpurun_out/r06_g_bench.txt
done
DDIF_OP_TIMING=$R/gpurun_out/r06_g_op_timing.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
grep -E "linattn" gpurun_out/r06_g_op_timing.csv | cut -d, -f2- | sort | uniq -c | head -20
