#!/bin/bash
# round 6, call d: the 8 x 8 decoder attention half as one launch (kernels_lafuse8.h) -- parity + same-box A/B (DDIF_LA8=0 / 1 interleaved) + op table + small batch
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r06_d_tests.log
cat gpurun_out/r06_d_tests.log
for rep in 1 2 3; do
  for v in 0 1; do
    DDIF_LA8=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('LA8=$v rep $rep ms/step', round(r['ms_per_step']/200,4), 'launches', r['config']['launches_per_denoising_step'])" | tee -a gpurun_out/r06_d_la8_ab.txt
  done
done
DDIF_OP_TIMING=$R/gpurun_out/r06_d_op_timing.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
grep -E "linattn8|attn_block" gpurun_out/r06_d_op_timing.csv | head -8
for b in 8 16; do
  for v in 0 1; do
    DDIF_LA8=$v python3 bench.py --config gf2_dpm50 --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-shares 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('gf2 tiles=$b LA8=$v ms/job', round(r['ms_per_step'],2))" | tee -a gpurun_out/r06_d_la8_ab.txt
  done
done
