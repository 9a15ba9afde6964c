#!/bin/bash
# round 6, call l: 8 x 16 tiles where 16 x 16 ones leave CUs idle (half-tile statistics partials keep batch bit-equality; DDIF_TILE16) -- parity + batch tests, A/B at 8 / 16 / 64 tiles
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch64.py -m gpu -x -q 2>&1 | tail -3) > gpurun_out/r06_l_tests.log
cat gpurun_out/r06_l_tests.log
for b in 8 16 32; do
  for v in 1 -; do
    if [ $v = 1 ]; then export DDIF_TILE16=1; else unset DDIF_TILE16; fi
    python3 bench.py --config gf2_dpm50 --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-shares 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('gf2 tiles=$b TILE16=$v (- = the plan chooses) ms/job', round(r['ms_per_step'],2))" | tee -a gpurun_out/r06_l_tile_ab.txt
  done
done
unset DDIF_TILE16
for rep in 1 2; do
  for v in 1 -; do
    if [ $v = 1 ]; then export DDIF_TILE16=1; else unset DDIF_TILE16; fi
    python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('wv3 B=64 TILE16=$v rep $rep ms/step', round(r['ms_per_step']/200,4))" | tee -a gpurun_out/r06_l_tile_ab.txt
  done
done
