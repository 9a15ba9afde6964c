#!/bin/bash
# usage: tools/gpu_graph_ab.sh <tag>   -- hipGraph replay vs plain stream launches at B = 64: job time and the device copies
# (__amd_rocclr_copyBuffer) each mode puts into the kernel trace
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  cd $R
  DDIF_GRAPH=$g python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > $R/gpurun_out/${tag}_bench_graph$g.json 2> $R/gpurun_out/${tag}_bench_graph$g.log
  python3 - <<PY
import json
r=json.load(open("$R/gpurun_out/${tag}_bench_graph$g.json"))
print("GRAPH=$g ms/denoise-step", r["ms_per_step"]/r["config"]["T"])
PY
  export DDIF_GRAPH=$g
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${tag}_kt_graph$g -- python3 bench.py --steps 1 --warmup 1 --T 20 --no-cpu-baseline > /dev/null 2>&1
  unset DDIF_GRAPH
  f=$(find $R/gpurun_out/${tag}_kt_graph$g -name '*kernel_stats.csv' | head -1)
  cp "$f" $R/gpurun_out/${tag}_kernel_stats_graph$g.csv
  grep -i "rocclr" $R/gpurun_out/${tag}_kernel_stats_graph$g.csv
  rm -rf $R/gpurun_out/${tag}_kt_graph$g
done
