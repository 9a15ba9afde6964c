#!/bin/bash
# usage: tools/gpu_quick.sh <tag> [extra bench args]   -- gpu tests + T=20 bench with rocprof kernel stats
tag=$1; shift
mkdir -p gpurun_out
(python -m pytest tests -m gpu -q -x 2>&1 | tail -6) > gpurun_out/${tag}_tests.log 2>&1
cat gpurun_out/${tag}_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --T 20 --no-cpu-baseline "$@" > $GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json 2> /tmp/prof_$tag.log
grep -E "bench|Error|error" /tmp/prof_$tag.log | tail -5
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv") $GRAFT_REPO_ROOT/gpurun_out/${tag}_kernel_stats.csv
python3 - <<PY
import json
r=json.load(open("$GRAFT_REPO_ROOT/gpurun_out/${tag}_bench.json"))
print("ms/denoise-step", r["ms_per_step"]/r["config"]["T"], "conv3x3 TF", r["roofline"]["achieved"], "job TF", r["roofline"]["whole_job_tflops"])
PY
