#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for n in 2 6; do
  rm -rf /tmp/pt_$n
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pt_$n -o p -- python3 $R/bench.py --config wv3_train_b32 --steps $n --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pt_$n.log
  f=$(find /tmp/pt_$n -name "*kernel_stats.csv")
  echo "== steps $n"
  python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(int(r['Calls']) for r in rows)
print('total launches', tot)
for r in rows:
    if any(k in r['Name'] for k in ('copyBuffer','FillFunctor','fillBuffer','wgrad_kernel<1, 0>','elementwise','CatArray','index')):
        print(r['Name'][:90], r['Calls'])
P
done
