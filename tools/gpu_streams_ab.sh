#!/bin/bash
# usage: tools/gpu_streams_ab.sh <tag> -- the two-stream reverse pass: its tests, then the config-5 bench line with DDIF_TRAIN_STREAMS=0 / 1, same box
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests/test_train_graph.py -m gpu -q -x 2>&1 | tail -6) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
for v in 0 1 0 1; do
  DDIF_TRAIN_STREAMS=$v python3 bench.py --config wv3_train_b32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_streams$v.json 2> gpurun_out/${tag}_streams$v.log
  python3 -c "
import json; r=json.load(open('gpurun_out/${tag}_streams$v.json')); print('DDIF_TRAIN_STREAMS=$v', r['value'], r['unit'], 'ms/iter', r['ms_per_step'], 'sc passes', r['config'].get('self_conditioning_passes_in_timed_region'))" | tee -a gpurun_out/${tag}_streams_ab.txt
done
