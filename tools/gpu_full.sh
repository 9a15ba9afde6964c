#!/bin/bash
# usage: tools/gpu_full.sh <tag>   -- everything a round's profiles/ needs, in one gpurun call:
#   gpu tests, rocprofv3 kernel stats (T=20), two PMC passes (FETCH_SIZE, WRITE_SIZE; T=4), the default bench (T=1000)
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests -m gpu -q -x 2>&1 | tail -6) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/bench.py --steps 1 --warmup 1 --T 20 --no-cpu-baseline > $R/gpurun_out/${tag}_bench_T20.json 2> /tmp/prof_$tag.log
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv") $R/gpurun_out/${tag}_kernel_stats_T20_B64.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --steps 1 --warmup 0 --T 4 --no-cpu-baseline > /dev/null 2> /tmp/pmc_$c.log
  python3 $R/tools/pmc_summary.py $(find /tmp/pmc_$c -name "*counter_collection.csv") 60 > $R/gpurun_out/${tag}_pmc_$c.csv
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/${tag}_pmc_FETCH_SIZE.csv $R/gpurun_out/${tag}_pmc_WRITE_SIZE.csv $R/gpurun_out/${tag}_hbm_traffic.json
cd $R
python3 bench.py > gpurun_out/${tag}_bench_T1000_B64.json 2> gpurun_out/${tag}_bench_T1000_B64.log
cat gpurun_out/${tag}_bench_T1000_B64.json
