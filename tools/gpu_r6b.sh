#!/bin/bash
# round 6, call b: attn_block on 8 waves -- parity slice + same-box A/B (DDIF_ATTN_NW=4 / 8 interleaved)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_forward_matches_reference_golden or test_forward_matches_oracle_other_sizes" 2>&1 | tail -3) > gpurun_out/r06_b_tests.log
cat gpurun_out/r06_b_tests.log
for rep in 1 2; do
  for nw in 4 8; do
    DDIF_ATTN_NW=$nw python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('ATTN_NW=$nw rep $rep ms/step', round(r['ms_per_step']/200,4))" | tee -a gpurun_out/r06_b_attn_nw_ab.txt
  done
done
DDIF_OP_TIMING=$R/gpurun_out/r06_b_op_timing.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
grep attn_block gpurun_out/r06_b_op_timing.csv | head -3
