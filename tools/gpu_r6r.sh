#!/bin/bash
# round 6, call r: row staging of the low-resolution 3x3 convs (kernels_lr.h ROWS) -- parity slice first, then DDIF_LR_ROWS=0 / default interleaved at B = 64,
# then the per-op table of the low-resolution class either way
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_forward_matches_reference_golden or test_ddpm_matches_reference_golden and ddpm_wv3_16_T10 or test_forward_matches_oracle_other_sizes" -p no:cacheprovider 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_env_switches.py -m gpu -x -q -k "resident_weights or LR_ROWS" -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2 3; do
  for v in 0 -; do
    if [ $v = 0 ]; then export DDIF_LR_ROWS=0; else unset DDIF_LR_ROWS; fi
    python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('LR_ROWS=$v rep $rep ms/step', round(r['ms_per_step']/200,4))"
  done
done
for v in 0 -; do
  if [ $v = 0 ]; then export DDIF_LR_ROWS=0; else unset DDIF_LR_ROWS; fi
  DDIF_OP_TIMING=$GRAFT_REPO_ROOT/gpurun_out/r06_r_ops_rows$v.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
  python3 - <<P
import csv
rows=list(csv.DictReader(open("gpurun_out/r06_r_ops_rows$v.csv")))
lr=[r for r in rows if r["kernel"].startswith("lr3x3")]
print("LR_ROWS=$v lr3x3 launches", len(lr), "sum us", round(sum(float(r["us"]) for r in lr),1), "all ops us", round(sum(float(r["us"]) for r in rows),1))
P
done
