#!/bin/bash
# round 5, call q: per-kernel durations of the 8-tile GF2 share (50 NFE), round 4's tree against this tree on one box (rocprofv3 --kernel-trace --stats)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for t in r04 r05; do
  rm -rf /tmp/q_$t
  if [ $t = r04 ]; then D=$R/_r04; else D=$R; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/q_$t -o p -- python3 $D/bench.py --config gf2_dpm50 --batch 8 --steps 2 --warmup 1 --no-cpu-baseline > /tmp/q_$t.json 2> /tmp/q_$t.log
  cp $(find /tmp/q_$t -name "*kernel_stats.csv") $R/gpurun_out/r05_q_kstats_$t.csv
  python3 -c "import json; r=json.load(open('/tmp/q_$t.json')); print('$t ms/job (profiled)', r['ms_per_step'])"
done
python3 - <<PY
import csv, re
def load(p):
    d = {}
    for r in csv.DictReader(open(p)):
        n = re.sub(r"\(.*", "", r["Name"])[:80]
        d[n] = (int(r["Calls"]), float(r["TotalDurationNs"]) / 1e3)
    return d
a, b = load("$R/gpurun_out/r05_q_kstats_r04.csv"), load("$R/gpurun_out/r05_q_kstats_r05.csv")
ta, tb = sum(v[1] for v in a.values()), sum(v[1] for v in b.values())
print("total kernel us: r04 %.0f  r05 %.0f" % (ta, tb))
for n in sorted(set(a) | set(b), key=lambda n: -(b.get(n, (0, 0))[1] + a.get(n, (0, 0))[1]))[:34]:
    ca, ua = a.get(n, (0, 0.0)); cb, ub = b.get(n, (0, 0.0))
    print("%-82s r04 %5d x %6.1f = %8.0f   r05 %5d x %6.1f = %8.0f" % (n, ca, ua / max(1, ca), ua, cb, ub / max(1, cb), ub))
PY
