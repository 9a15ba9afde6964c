#!/bin/bash
# usage: tools/gpu_train_ab.sh <tag> VAR v1 v2 ...   -- same-box A/B of one environment switch on the training bench (config 5)
tag=$1; var=$2; shift; shift
mkdir -p gpurun_out
for v in "$@"; do
  env $var=$v python3 bench.py --config wv3_train_b32 --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${tag}_${var}_$v.json 2> gpurun_out/${tag}_${var}_$v.log
  python3 -c "
import json; r=json.load(open('gpurun_out/${tag}_${var}_$v.json')); print('$var=$v', 'tiles/s %.1f' % r['value'], 'ms/iter %.2f' % r['ms_per_step'])"
done
