#!/bin/bash
# round 6, call v: ONE persistent workgroup per CU for every conv launch (DDIF_BENCH_GRID_CAP=256 through the test hook) against the plan's two: per-op table either way.
# Question: the 3x3 kernels of the 64x64 / 32x32 levels execute a ~2 500-instruction prologue per workgroup for 2 work items -- do fewer, longer-lived workgroups pay?
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for v in - 256 384; do
  if [ $v = - ]; then unset DDIF_BENCH_GRID_CAP; else export DDIF_BENCH_GRID_CAP=$v; fi
  python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('cap=$v ms/step', round(r['ms_per_step']/200,4))"
  DDIF_OP_TIMING=$GRAFT_REPO_ROOT/gpurun_out/r06_v_ops_$v.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
done
python3 - <<P
import csv,collections
def load(v):
    agg=collections.OrderedDict()
    for r in csv.DictReader(open("gpurun_out/r06_v_ops_%s.csv"%v)):
        lvl=r["op"].split("@")[1].split()[0] if "@" in r["op"] else ""
        a=agg.setdefault((r["kernel"],lvl),[0,0.0]); a[0]+=1; a[1]+=float(r["us"])
    return agg
a,b,c=load("-"),load("256"),load("384")
for k in a:
    print("%-28s %-7s n=%2d  default %7.1f  cap256 %7.1f  cap384 %7.1f"%(k[0],k[1],a[k][0],a[k][1],b.get(k,[0,0])[1],c.get(k,[0,0])[1]))
P
