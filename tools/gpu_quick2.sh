#!/bin/bash
# usage: tools/gpu_quick2.sh <tag> [pytest -k expr]  -- op-by-op timing (T=40), a T=200 bench with the class breakdown, and an optional parity slice
tag=$1; R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
if [ -n "$2" ]; then python -m pytest tests -m gpu -q -x -k "$2" 2>&1 | tail -3; fi
DDIF_OP_TIMING=$R/gpurun_out/${tag}_op_timing.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > $R/gpurun_out/${tag}_bench_T200.json 2>/dev/null
python3 - <<PY
import json, csv
r=json.load(open("$R/gpurun_out/${tag}_bench_T200.json"))
print("ms/step", round(r["ms_per_step"]/200,3), [(c["class"][:8], round(c["ms_per_step"],3), c["launches_per_step"]) for c in r["roofline"]["whole_step"]["classes"]])
rows=list(csv.DictReader(open("$R/gpurun_out/${tag}_op_timing.csv")))
seen={}
for x in rows:
    k=x["kernel"]
    seen.setdefault(k,[0,0.0]); seen[k][0]+=1; seen[k][1]+=float(x["us"])
for k,v in sorted(seen.items(), key=lambda kv:-kv[1][1])[:14]: print("  %-28s x%3d %8.1f us  avg %6.1f"%(k[:28],v[0],v[1],v[1]/v[0]))
PY
