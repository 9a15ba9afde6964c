#!/bin/bash
# round 6, call e: 192-channel 16 x 16 decoder block on linattn_fused (DDIF_LA6) + lafuse8 with both M_b rounds prefetched -- parity slice + A/B
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
./tools/mbench_la8.bin > gpurun_out/r06_e_mbench_la8.txt 2>&1; head -12 gpurun_out/r06_e_mbench_la8.txt
(python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward or ddpm_wv3_16_T10 or s10_o2" 2>&1 | tail -3) > gpurun_out/r06_e_tests.log
cat gpurun_out/r06_e_tests.log
for rep in 1 2 3; do
  for v in 0 1; do
    DDIF_LA6=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('LA6=$v rep $rep ms/step', round(r['ms_per_step']/200,4), 'launches', r['config']['launches_per_denoising_step'])" | tee -a gpurun_out/r06_e_la6_ab.txt
  done
done
DDIF_OP_TIMING=$R/gpurun_out/r06_e_op_timing.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
grep -E "linattn" gpurun_out/r06_e_op_timing.csv | head -16
