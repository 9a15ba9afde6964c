#!/bin/bash
# round 5, call o: op-by-op timing of one profiled step under DDIF_XCD=0 and 15 on ONE box (which launches gain, which lose)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2; do for v in 0 15; do
DDIF_XCD=$v DDIF_OP_TIMING=$R/gpurun_out/r05_o_op_xcd${v}_$rep.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
done; done
python3 - <<PY
import csv
def load(v):
    rows=[list(csv.DictReader(open("gpurun_out/r05_o_op_xcd%d_%d.csv" % (v, r)))) for r in (1, 2)]
    return [(a["op"], a["kernel"], min(float(a["us"]), float(b["us"]))) for a, b in zip(*rows)]
a, b = load(0), load(15)
ta = tb = 0
for (op, k, ua), (_, _, ub) in zip(a, b):
    ta += ua; tb += ub
    if abs(ub - ua) > 0.06 * ua: print("%-66s %-20s %6.1f -> %6.1f" % (op[:66], k[:20], ua, ub))
print("sum %.1f -> %.1f" % (ta, tb))
PY
