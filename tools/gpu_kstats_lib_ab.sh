#!/bin/bash
# usage: tools/gpu_kstats_lib_ab.sh <regex>  -- per-kernel durations (rocprofv3 --kernel-trace --stats) of the training bench under lib/libddif_old.so and lib/libddif.so, lines matching <regex>
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in old new; do
  rm -rf /tmp/g_$v
  if [ $v = old ]; then
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g_$v -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 4 --warmup 2 --no-cpu-baseline --lib $R/dif-pan_amd/lib/libddif_old.so > /dev/null 2> /tmp/g_$v.log
  else
    timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/g_$v -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 4 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/g_$v.log
  fi
  f=$(find /tmp/g_$v -name "*kernel_stats.csv" | head -1)
  if [ -z "$f" ]; then tail -3 /tmp/g_$v.log; continue; fi
  python3 - "$f" "$1" "$v" <<PY
import csv, re, sys
for r in csv.DictReader(open(sys.argv[1])):
    if re.search(sys.argv[2], r["Name"]):
        print("%-4s %-60s %5d x %7.1f us" % (sys.argv[3], re.sub(r"\\(.*", "", r["Name"])[:60], int(r["Calls"]), float(r["TotalDurationNs"]) / int(r["Calls"]) / 1e3))
PY
done
