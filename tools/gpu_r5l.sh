#!/bin/bash
# round 5, call l: L2 hit rate per kernel with and without the XCD-contiguous partition (rocprofv3 PMC TCC_HIT_sum TCC_MISS_sum, T = 4)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for v in 0 15; do
  rm -rf /tmp/pmc_l2_$v
  export DDIF_XCD=$v
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d /tmp/pmc_l2_$v -o p -- python3 $R/bench.py --steps 1 --warmup 0 --T 4 --no-cpu-baseline > /dev/null 2> /tmp/pmc_l2_$v.log
  f=$(find /tmp/pmc_l2_$v -name "*counter_collection.csv")
  python3 $R/tools/pmc_summary.py $f 40 > $R/gpurun_out/r05_l_l2_xcd$v.csv
  echo "== DDIF_XCD=$v"; head -30 $R/gpurun_out/r05_l_l2_xcd$v.csv
done
