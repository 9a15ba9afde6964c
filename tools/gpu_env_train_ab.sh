#!/bin/bash
# usage: tools/gpu_env_train_ab.sh <tag> <ENV_VAR> -- the config-5 bench line with <ENV_VAR> unset / set to 1, alternating, same box (after the train tests)
tag=$1; var=$2
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests/test_train_graph.py -m gpu -q -x 2>&1 | tail -6) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
for v in "" 1 "" 1; do
  if [ -n "$v" ]; then export $var=$v; else unset $var; fi
  python3 bench.py --config wv3_train_b32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_ab.json 2> gpurun_out/${tag}_ab.log
  python3 -c "
import json; r=json.load(open('gpurun_out/${tag}_ab.json')); print('$var=$v', r['value'], r['unit'], 'ms/iter', r['ms_per_step'])" | tee -a gpurun_out/${tag}_${var}_ab.txt
done
