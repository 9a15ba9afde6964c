#!/bin/bash
# round 5, call s: the bf16x3 weight-gradient kernel (conv3x3_wgrad_x3_kernel) -- training parity tests, then the training bench with DDIF_WGRAD_X3=0 / 1 interleaved,
# then per-kernel stats of both
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_train_graph.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3; do for v in 0 1; do
  DDIF_WGRAD_X3=$v python3 bench.py --config wv3_train_b32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.load(sys.stdin); print('DDIF_WGRAD_X3=$v training ms/iteration %.2f' % r['ms_per_step'])"
done; done
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  rm -rf /tmp/s_$v
  DDIF_WGRAD_X3=$v rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/s_$v -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 6 --warmup 2 --no-cpu-baseline > /tmp/s_$v.json 2> /tmp/s_$v.log
  cp $(find /tmp/s_$v -name "*kernel_stats.csv") $R/gpurun_out/r05_s_train_kstats_x3_$v.csv
  python3 - <<PY
import csv, re
rows = list(csv.DictReader(open("$R/gpurun_out/r05_s_train_kstats_x3_$v.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("DDIF_WGRAD_X3=$v total kernel ms %.1f" % (tot / 1e6))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = re.sub(r"\(.*", "", r["Name"])
    if "wgrad" in n: print("   %-60s %5d x %7.1f us = %7.2f ms" % (n[:60], int(r["Calls"]), float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
