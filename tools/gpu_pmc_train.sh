#!/bin/bash
# usage: tools/gpu_pmc_train.sh <tag> "<counters pass 1>" ["<counters pass 2>" ...] -- PMC counters per kernel over two training iterations (one rocprofv3 pass per argument)
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
i=0
for c in "$@"; do
  i=$((i+1))
  rm -rf /tmp/pmct_$i
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmct_$i -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> /tmp/pmct_$i.log
  f=$(find /tmp/pmct_$i -name "*counter_collection.csv")
  if [ -n "$f" ]; then python3 $R/tools/pmc_summary.py $f 40 > $R/gpurun_out/${tag}_pmc_train_$i.csv; head -14 $R/gpurun_out/${tag}_pmc_train_$i.csv | cut -c1-220; else tail -5 /tmp/pmct_$i.log; fi
done
