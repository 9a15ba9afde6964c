#!/bin/bash
# usage: tools/gpu_timeline.sh <tag> -- one training iteration's kernel timeline: wall time, busy time per queue, time with two kernels in flight, idle gaps
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl_$tag -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 3 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/tl_$tag.log
python3 - <<PY | tee $R/gpurun_out/${tag}_timeline.txt
import csv, glob, collections
f = glob.glob("/tmp/tl_$tag/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
print("columns:", [c for c in rows[0].keys()][:16])
qk = "Queue_Id" if "Queue_Id" in rows[0] else ("Stream_Id" if "Stream_Id" in rows[0] else None)
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qk] if qk else "?", r["Kernel_Name"]) for r in rows]
ev.sort()
# the last iteration: between the last two refresh_blob_kernel launches
ref = [e for e in ev if "refresh_blob_kernel" in e[3]]
t0, t1 = ref[-2][0], ref[-1][0]
it = [e for e in ev if t0 <= e[0] < t1]
print("iteration wall ms", (t1 - t0) / 1e6, "kernels", len(it))
byq = collections.defaultdict(float)
for s, e, q, n in it: byq[q] += (e - s) / 1e6
print("busy ms per queue:", dict(byq))
# sweep
pts = []
for s, e, q, n in it: pts += [(s, 1), (e, -1)]
pts.sort()
act = 0; last = t0; hist = collections.defaultdict(float)
for t, d in pts:
    hist[min(act, 3)] += (t - last) / 1e6
    act += d; last = t
hist[min(act, 3)] += (t1 - last) / 1e6
print("ms with N kernels in flight:", dict(sorted(hist.items())))
# what runs while the side queue is busy / idle on main
PY
