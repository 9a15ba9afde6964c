#!/bin/bash
# usage: tools/gpu_kernel_trace.sh <tag> <kernel-name substring> -- per-launch durations (grid, LDS) of the matching kernels over ONE training iteration, in launch order
tag=$1; pat=$2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/tr_$tag.log
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/tr_$tag/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "$pat" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
agg = collections.OrderedDict()
for r in rows[len(rows) // 2:]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    key = (r["Kernel_Name"].split("(")[0], r.get("Grid_Size_X", r.get("Grid_Size", "?")), r.get("Grid_Size_Y", ""), r.get("LDS_Block_Size", r.get("LDS_Block_Size_v", "")))
    a = agg.setdefault(key, [0, 0.0])
    a[0] += 1; a[1] += us
out = open("$R/gpurun_out/${tag}_trace_${pat}.txt", "w")
for k, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    line = "%-40s grid %8s x %-4s lds %7s  x%-3d total %8.1f us  avg %7.1f us" % (k[0], k[1], k[2], k[3], c, us, us / c)
    print(line); out.write(line + "\n")
PY
