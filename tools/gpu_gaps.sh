#!/bin/bash
# usage: tools/gpu_gaps.sh <tag> [env assignments...]  -- kernel trace of a T=6 job; prints inter-kernel gap statistics
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for e in "$@"; do export "$e"; done
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o p -- python3 $R/bench.py --steps 1 --warmup 1 --T 6 --no-cpu-baseline > /dev/null 2> /tmp/tr_$tag.log
python3 - <<PY
import csv, glob
f = glob.glob("/tmp/tr_$tag/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows]
# last job only: take the last 6 steps worth -> use the last 1200 kernels
ks = ks[-1200:]
gaps = [(ks[i + 1][0] - ks[i][1]) / 1e3 for i in range(len(ks) - 1)]
dur = [(k[1] - k[0]) / 1e3 for k in ks]
span = (ks[-1][1] - ks[0][0]) / 1e3
import statistics as st
g = sorted(gaps)
print("kernels %d  span %.1f us  sum(dur) %.1f us  sum(gaps) %.1f us (%.1f%%)  median gap %.2f  p90 %.2f  mean %.2f" % (len(ks), span, sum(dur), sum(gaps), 100 * sum(gaps) / span, g[len(g) // 2], g[int(len(g) * 0.9)], sum(gaps) / len(gaps)))
neg = sum(1 for x in gaps if x < 0)
print("overlapping (negative gap):", neg)
PY
