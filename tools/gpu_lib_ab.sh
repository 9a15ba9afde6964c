#!/bin/bash
# usage: tools/gpu_lib_ab.sh <tag> <other-lib.so>  -- parity tests on the default build, then a same-box A/B of the default build against another build
# of libddif (bench.py --lib): T = 200 job time twice each, alternating, and one op-timing run each
tag=$1; other=$2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch64.py -m gpu -q -x -k "not T1000" 2>&1 | tail -4) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
for v in new old new old; do
  L=""; [ $v = old ] && L="--lib $other"
  python3 bench.py $L --steps 2 --warmup 1 --T 200 --no-cpu-baseline > $R/gpurun_out/${tag}_bench_$v.json 2> $R/gpurun_out/${tag}_bench_$v.log
  python3 - <<PY
import json
r=json.load(open("$R/gpurun_out/${tag}_bench_$v.json"))
c={x["class"][:12]:round(x["ms_per_step"],3) for x in r["roofline"]["whole_step"]["classes"]}
print("$v ms/denoise-step", round(r["ms_per_step"]/r["config"]["T"],4), c)
PY
done
