#!/bin/bash
# usage: tools/gpu_train.sh <tag>  -- the training path on the GPU: its tests, the config-5 bench line, a kernel-level profile of a few iterations
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests/test_train_graph.py tests/test_backward_ops.py tests/test_aux_kernels.py -m gpu -q -x 2>&1 | tail -12) > $R/gpurun_out/${tag}_train_tests.log 2>&1
cat $R/gpurun_out/${tag}_train_tests.log
python3 bench.py --config wv3_train_b32 --steps 10 --warmup 3 --cpu-seconds 12 > gpurun_out/${tag}_bench_train.json 2> gpurun_out/${tag}_bench_train.log
tail -3 gpurun_out/${tag}_bench_train.log
python3 -c "
import json; r=json.load(open('gpurun_out/${tag}_bench_train.json')); print('train', r['value'], r['unit'], 'ms/iter', r['ms_per_step'], 'frac', r['roofline']['frac'], 'cpu', r.get('cpu_baseline',{}).get('value'))"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/prof_$tag.log
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv") $R/gpurun_out/${tag}_train_kernel_stats.csv
head -25 $R/gpurun_out/${tag}_train_kernel_stats.csv | cut -c1-150
