#!/bin/bash
# usage: tools/gpu_quick3.sh <tag> [T]  -- all GPU tests, a parity-margin report, then a T-step bench and one op-timing run of the default build
tag=$1; T=${2:-200}
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(timeout 2400 python -m pytest tests -m gpu -q -x 2>&1 | tail -6) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
python3 tools/parity_report.py > $R/gpurun_out/${tag}_parity_report.txt 2>&1; tail -12 $R/gpurun_out/${tag}_parity_report.txt
for k in 1 2; do
python3 bench.py --steps 2 --warmup 1 --T $T --no-cpu-baseline > $R/gpurun_out/${tag}_bench_T$T.json 2> $R/gpurun_out/${tag}_bench_T$T.log
python3 - <<PY
import json
r=json.load(open("$R/gpurun_out/${tag}_bench_T$T.json"))
c={x["class"][:12]:(x["launches_per_step"], round(x["ms_per_step"],3)) for x in r["roofline"]["whole_step"]["classes"]}
print("ms/denoise-step", round(r["ms_per_step"]/r["config"]["T"],4), c)
PY
done
DDIF_OP_TIMING=$R/gpurun_out/${tag}_op_timing.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
