#!/bin/bash
# round 6, call k: attn_block on FOUR workgroups per sample (DDIF_ATTN_SPLIT=4) against two -- microbenchmark, parity slice, same-box A/B
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
./tools/mbench_attn.bin > gpurun_out/r06_k_mbench_attn.txt 2>&1; grep "per launch" gpurun_out/r06_k_mbench_attn.txt
(DDIF_ATTN_SPLIT=4 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward or ddpm_wv3_64 or dpm_gf2_64" 2>&1 | tail -3) > gpurun_out/r06_k_tests.log
cat gpurun_out/r06_k_tests.log
for rep in 1 2 3; do
  for v in 2 4; do
    DDIF_ATTN_SPLIT=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('ATTN_SPLIT=$v rep $rep ms/step', round(r['ms_per_step']/200,4))" | tee -a gpurun_out/r06_k_attn_split4_ab.txt
  done
done
