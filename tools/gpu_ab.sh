#!/bin/bash
# usage: tools/gpu_ab.sh <tag> [T]   -- gpu tests, then op-by-op timing (DDIF_OP_TIMING, one step) and a T-step bench for
# the default build and for DDIF_LR=0 (A/B of the low-resolution kernel)
tag=$1; T=${2:-200}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
for lr in 1 0; do
  DDIF_LR=$lr DDIF_OP_TIMING=$R/gpurun_out/${tag}_op_timing_lr$lr.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2> $R/gpurun_out/${tag}_optiming_lr$lr.log
  DDIF_LR=$lr python3 bench.py --steps 2 --warmup 1 --T $T --no-cpu-baseline > $R/gpurun_out/${tag}_bench_T${T}_lr$lr.json 2> $R/gpurun_out/${tag}_bench_lr$lr.log
  python3 - <<PY
import json
r=json.load(open("$R/gpurun_out/${tag}_bench_T${T}_lr$lr.json"))
print("LR=$lr ms/denoise-step", r["ms_per_step"]/r["config"]["T"], "job TF", r["roofline"]["whole_job_tflops"])
PY
done
