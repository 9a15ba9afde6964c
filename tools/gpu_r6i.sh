#!/bin/bash
# round 6, call i: linattn_fused on four-wave workgroups where eight-wave ones leave CUs idle (DDIF_LA_NW) -- parity slice, same-box A/B at B = 64 and at 8 / 16 GF2 tiles
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward or ddpm_wv3_16_T10 or s10_o2" 2>&1 | tail -3) > gpurun_out/r06_i_tests.log
cat gpurun_out/r06_i_tests.log
(DDIF_LA_NW=4 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "forward" 2>&1 | tail -3) >> gpurun_out/r06_i_tests.log
tail -2 gpurun_out/r06_i_tests.log
for rep in 1 2 3; do
  for v in 8 0; do
    DDIF_LA_NW=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('LA_NW=$v (0 = default choice) rep $rep ms/step', round(r['ms_per_step']/200,4))" | tee -a gpurun_out/r06_i_la_nw_ab.txt
  done
done
for b in 8 16; do
  for v in 8 0; do
    DDIF_LA_NW=$v python3 bench.py --config gf2_dpm50 --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-shares 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('gf2 tiles=$b LA_NW=$v ms/job', round(r['ms_per_step'],2))" | tee -a gpurun_out/r06_i_la_nw_ab.txt
  done
done
DDIF_OP_TIMING=$R/gpurun_out/r06_i_op_timing.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
grep -E "linattn_fused" gpurun_out/r06_i_op_timing.csv | grep "16x16" | cut -d, -f2-
