#!/bin/bash
# round 6, call o: attn_block's qkv conv on f16x2 (DDIF_ATTN_F16) -- parity slice + same-box A/B
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(python -m pytest tests/test_gpu_parity.py tests/test_range_guard.py -m gpu -x -q -k "forward or ddpm_wv3_64 or dpm_gf2_64 or trained_like" 2>&1 | tail -3) > gpurun_out/r06_o_tests.log
cat gpurun_out/r06_o_tests.log
for rep in 1 2 3; do
  for v in 0 1; do
    DDIF_ATTN_F16=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('ATTN_F16=$v rep $rep ms/step', round(r['ms_per_step']/200,4))" | tee -a gpurun_out/r06_o_attn_f16_ab.txt
  done
done
python3 tools/parity_report.py 2>/dev/null | grep -E "fwd_wv3_64|ddpm_wv3_64_T1000 |dpm_gf2_64" | tee -a gpurun_out/r06_o_attn_f16_ab.txt
