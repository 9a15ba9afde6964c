#!/bin/bash
# round 6, call j: attn_block on two workgroups per sample (token halves; DDIF_ATTN_SPLIT) -- microbenchmark, parity slice, same-box A/B
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
./tools/mbench_attn.bin > gpurun_out/r06_j_mbench_attn.txt 2>&1; grep "per launch" gpurun_out/r06_j_mbench_attn.txt
(python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward or ddpm_wv3_64 or dpm_gf2_64" 2>&1 | tail -3) > gpurun_out/r06_j_tests.log
cat gpurun_out/r06_j_tests.log
for rep in 1 2 3; do
  for v in 1 2; do
    DDIF_ATTN_SPLIT=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('ATTN_SPLIT=$v rep $rep ms/step', round(r['ms_per_step']/200,4))" | tee -a gpurun_out/r06_j_attn_split_ab.txt
  done
done
for b in 8; do
  for v in 1 2; do
    DDIF_ATTN_SPLIT=$v python3 bench.py --config gf2_dpm50 --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-shares 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('gf2 tiles=$b ATTN_SPLIT=$v ms/job', round(r['ms_per_step'],2))" | tee -a gpurun_out/r06_j_attn_split_ab.txt
  done
done
