#!/bin/bash
# usage: tools/gpu_mix_ab.sh <tag>   -- parity tests, then job time and per-op timing with and without the epilogue-fused ffn.3
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch64.py tests/test_env_switches.py -m gpu -q -x -k "not T1000" 2>&1 | tail -6) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
for m in 1 0 1 0; do
  DDIF_MIX=$m python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > $R/gpurun_out/${tag}_bench_mix$m.json 2> $R/gpurun_out/${tag}_bench_mix$m.log
  python3 - <<PY
import json
r=json.load(open("$R/gpurun_out/${tag}_bench_mix$m.json"))
print("MIX=$m ms/denoise-step", r["ms_per_step"]/r["config"]["T"])
PY
done
for m in 1 0; do
  DDIF_MIX=$m DDIF_OP_TIMING=$R/gpurun_out/${tag}_op_timing_mix$m.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
  grep -i "ffn.2\|ffn.3" $R/gpurun_out/${tag}_op_timing_mix$m.csv | grep "@64x64" | head -4
done
