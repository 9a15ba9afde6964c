#!/bin/bash
# usage: tools/gpu_train_idle.sh <tag>  -- is the training iteration ever waiting for the host?  Kernel trace of a short training job: per iteration the wall span,
# the union of the kernels' busy intervals (both streams) and the idle time in between
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ti_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/ti_$tag -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 4 --warmup 2 --no-cpu-baseline > /tmp/ti_$tag.json 2> /tmp/ti_$tag.log
python3 - <<PY
import csv, glob, json
f = glob.glob("/tmp/ti_$tag/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
print("ms per iteration (bench, under the profiler):", json.load(open("/tmp/ti_$tag.json"))["ms_per_step"])
# iterations: split at the optimizer's fused step kernel (one per iteration)
marks = [i for i, r in enumerate(rows) if "optim_update_kernel" in r[2]]
print("kernels", len(rows), "iteration marks", len(marks))
names = sorted(set(r[2][:40] for r in rows if "ddif" in r[2]))
if len(marks) >= 3:
    for a, b in zip(marks[-3:-1], marks[-2:]):
        seg = rows[a + 1:b + 1]
        t0, t1 = seg[0][0], max(r[1] for r in seg)
        busy, cur_s, cur_e = 0, seg[0][0], seg[0][1]
        gaps = []
        for s, e, _ in seg[1:]:
            if s > cur_e:
                busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        big = [g for g in gaps if g > 5000]
        # the ten largest gaps with the kernels on either side
        ends = sorted(seg, key=lambda r: r[1])
        gl = []
        cur_e, last = seg[0][1], seg[0][2]
        for s, e, n in seg[1:]:
            if s > cur_e: gl.append((s - cur_e, last, n))
            if e > cur_e: cur_e, last = e, n
        for g, a_, b_ in sorted(gl, reverse=True)[:10]: print("      gap %8.1f us   after %-50s before %-50s" % (g / 1e3, a_[:50], b_[:50]))
        print("iteration: %d kernels, span %.2f ms, busy (union) %.2f ms, idle %.2f ms in %d gaps (%d gaps > 5 us = %.2f ms, largest %.1f us), sum of kernel durations %.2f ms" % (
            len(seg), (t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(gaps), len(big), sum(big) / 1e6, max(gaps) / 1e3 if gaps else 0, sum(e - s for s, e, _ in seg) / 1e6))
else:
    print([r[2][:60] for r in rows[-40:]])
PY
