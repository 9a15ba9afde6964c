#!/bin/bash
# usage: tools/gpu_wgrad_trace.sh <tag>  -- per-launch durations of the weight-gradient kernel matched with each launch's geometry
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
DDIF_WGRAD_DUMP=1 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/tr_$tag.log
grep "^\[wgrad\]" /tmp/tr_$tag.log > /tmp/geom_$tag.txt
python3 - <<PY
import csv, glob, collections
f = glob.glob("/tmp/tr_$tag/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "conv3x3_wgrad_kernel" in r["Kernel_Name"] or "conv3x3_wgrad_x3_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
geo = [l.strip() for l in open("/tmp/geom_$tag.txt")]
print("launches", len(rows), "geometry lines", len(geo))
n = len(rows) // 2  # two iterations (warmup + timed): use the second
agg = collections.OrderedDict()
for r, g in list(zip(rows, geo))[n:]:
    us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    a = agg.setdefault(g, [0, 0.0])
    a[0] += 1; a[1] += us
tot = 0
out = open("$R/gpurun_out/${tag}_wgrad_by_shape.txt", "w")
for g, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    kv = dict(t.split("=") for t in g.split()[1:])
    B, H, W, Ci, Co, ce = (int(kv[k]) for k in ("B", "H", "W", "Cin", "Cout", "centre"))
    fl = 2.0 * B * H * W * Ci * Co * (1 if ce else 9)
    ideal = fl / 157.3e12 * 1e6
    line = "%-110s x%-3d total %8.1f us  avg %7.1f us  mfma-bound %6.1f us  eff %.2f" % (g, c, us, us / c, ideal, ideal / (us / c))
    print(line); out.write(line + "\n"); tot += us
print("total wgrad us per iteration", tot); out.write("total %.1f\n" % tot)
PY
