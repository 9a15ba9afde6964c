#!/bin/bash
# usage: tools/gpu_split_ab.sh <tag>  -- same-box A/B of the forked regions (DDIF_SPLIT = sub-batches per region; DDIF_SPLIT_ALL = whole step)
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests/test_gpu_batch64.py tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -5) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
run() { # name, env...
  n=$1; shift
  env "$@" python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/${tag}_$n.json 2> gpurun_out/${tag}_$n.log
  python3 - <<PY
import json
try:
    r = json.load(open("gpurun_out/${tag}_$n.json"))
    print("$n", "ms/denoise %.3f" % r["roofline"]["whole_step"]["ms_per_denoising_step"], "MP/s %.5f" % r["value"], "arena MB", r["config"]["plan_memory_mb"]["step_activation_arena"])
except Exception as e:
    print("$n failed", e)
PY
}
run k1 DDIF_SPLIT=1
run k2 DDIF_SPLIT=2
run k4 DDIF_SPLIT=4
run k8 DDIF_SPLIT=8
run k16 DDIF_SPLIT=16
run all2 DDIF_SPLIT=2 DDIF_SPLIT_ALL=1
run all4 DDIF_SPLIT=4 DDIF_SPLIT_ALL=1
run k1b DDIF_SPLIT=1
for b in 8 16; do
  for k in 1 4; do
    DDIF_SPLIT=$k python3 bench.py --config gf2_dpm50 --batch $((b*b/b)) --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  done
done
