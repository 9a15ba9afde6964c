#!/bin/bash
# round 6, call u: the staged tile survives the work item (kernels_lr.h ROWS: exchange area of its own), so the next cout tile of the same (sample, tile) skips staging:
# parity slice + bit-equality, previous build against this one (B = 64 headline, per-op table), 8-tile GF2 share with DDIF_LR_ROWS=0 / default, whole round vs round 5's library
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_forward_matches_reference_golden or test_ddpm_matches_reference_golden and ddpm_wv3_16_T10 or test_ddim_matches_reference_golden and ddim_gf2 or test_forward_matches_oracle_other_sizes" -p no:cacheprovider 2>&1 | tail -2
timeout 900 python3 -m pytest tests/test_env_switches.py tests/test_gpu_batch64.py -m gpu -x -q -k "resident_weights or bit_equal or capped_grid" -p no:cacheprovider 2>&1 | tail -2
rm -f gpurun_out/r06_u_lib_ab.txt gpurun_out/r06_whole2_lib_ab.txt
bash tools/gpu_lib_ab2.sh r06_u dif-pan_amd/lib/libddif_prev.so 3
for which in other tree; do
  if [ $which = other ]; then L="--lib dif-pan_amd/lib/libddif_prev.so"; else L=""; fi
  DDIF_OP_TIMING=$GRAFT_REPO_ROOT/gpurun_out/r06_u_ops_$which.csv python3 bench.py $L --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
  python3 - <<P
import csv
rows=list(csv.DictReader(open("gpurun_out/r06_u_ops_$which.csv")))
for key in ("lr3x3_silu","lr3x3_gn_silu_rows","lr3x3_gn_silu_res_rows","lr3x3_res","lr3x3_gn_silu","lr1x1"):
    lr=[r for r in rows if r["kernel"].startswith(key)]
    if lr: print("$which", key, len(lr), "launches, sum us", round(sum(float(r["us"]) for r in lr),1))
lr=[r for r in rows if r["kernel"].startswith("lr3x3")]
print("$which lr3x3 all", len(lr), "sum us", round(sum(float(r["us"]) for r in lr),1), "| all ops us", round(sum(float(r["us"]) for r in rows),1))
P
done
for rep in 1 2; do
  for v in 0 -; do
    if [ $v = 0 ]; then export DDIF_LR_ROWS=0; else unset DDIF_LR_ROWS; fi
    python3 bench.py --config gf2_dpm50 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-shares 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('gf2 8 tiles LR_ROWS=$v rep $rep ms/job', round(r['ms_per_step'],2))"
  done
done
bash tools/gpu_lib_ab2.sh r06_whole2 dif-pan_amd/lib/libddif_r5.so 3
