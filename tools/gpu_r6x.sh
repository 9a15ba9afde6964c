#!/bin/bash
# round 6, call x: where the time of the 8-tile GF2 share goes (what one rank of an 8-GPU strong-scaling run holds): rocprofv3 kernel stats of one 50-NFE job of 8 tiles,
# and the per-op table at B = 8 (DDIF_OP_TIMING works on the DDPM loop: B = 8, T = 40)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_b8
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_b8 -o p -- python3 $R/bench.py --config gf2_dpm50 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline --no-shares > $R/gpurun_out/r06_x_bench_gf2_b8.json 2> /tmp/prof_b8.log
cp $(find /tmp/prof_b8 -name "*kernel_stats.csv") $R/gpurun_out/r06_x_kernel_stats_gf2_b8.csv
cd $R
DDIF_OP_TIMING=$R/gpurun_out/r06_x_op_timing_T40_B8.csv python3 bench.py --batch 8 --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
python3 bench.py --config gf2_dpm50 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline --no-shares 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('gf2 8 tiles ms/job', round(r['ms_per_step'],2))"
python3 - <<P
import csv,collections
rows=list(csv.DictReader(open("gpurun_out/r06_x_op_timing_T40_B8.csv")))
agg=collections.OrderedDict()
for r in rows:
    lvl=r["op"].split("@")[1].split()[0] if "@" in r["op"] else ""
    a=agg.setdefault((r["kernel"],lvl),[0,0.0]); a[0]+=1; a[1]+=float(r["us"])
tot=sum(v[1] for v in agg.values())
for k,v in sorted(agg.items(), key=lambda kv:-kv[1][1])[:24]:
    print("%-28s %-7s n=%2d sum %7.1f us avg %5.1f"%(k[0],k[1],v[0],v[1],v[1]/v[0]))
print("all ops us", round(tot,1), "launches", len(rows))
P
