#!/bin/bash
# usage: tools/gpu_lib_ab.sh <tag> <other lib (path relative to the repo)>  -- same-box A/B of the default library against another build of it:
# job time at T = 200 (twice each, interleaved) and one op-by-op timing of each
tag=$1; other=$2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2; do
  for v in default other; do
    if [ $v = other ]; then L="--lib $R/$other"; else L=""; fi
    python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline $L > gpurun_out/${tag}_${v}_$rep.json 2> /dev/null
    python3 -c "
import json; r=json.load(open('gpurun_out/${tag}_${v}_$rep.json')); print('$v', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
  done
done
DDIF_OP_TIMING=$R/gpurun_out/${tag}_op_default.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
DDIF_OP_TIMING=$R/gpurun_out/${tag}_op_other.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --lib $R/$other > /dev/null 2>&1
python3 - <<PY
import csv
a = list(csv.DictReader(open("gpurun_out/${tag}_op_default.csv"))); b = list(csv.DictReader(open("gpurun_out/${tag}_op_other.csv")))
tot = [0, 0]
for x, y in zip(a, b):
    ua, ub = float(x["us"]), float(y["us"]); tot[0] += ua; tot[1] += ub
    if abs(ua - ub) > 0.08 * ua: print("%-64s %-26s %7.1f -> %7.1f us" % (x["op"][:64], x["kernel"][:26], ua, ub))
print("sum of ops: %.1f -> %.1f us" % tuple(tot))
PY
