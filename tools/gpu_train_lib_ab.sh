#!/bin/bash
# usage: tools/gpu_train_lib_ab.sh <tag>  -- same-box A/B of two builds of the library (dif-pan_amd/lib/libddif_old.so against libddif.so) on the training bench,
# interleaved three times, preceded by the training parity tests on the new build
tag=$1
mkdir -p gpurun_out
(python -m pytest tests/test_train_graph.py tests/test_backward_ops.py -m gpu -q -x 2>&1 | tail -6) > gpurun_out/${tag}_tests.log 2>&1
cat gpurun_out/${tag}_tests.log
for rep in 1 2 3; do
  for v in old new; do
    if [ $v = old ]; then lib="--lib dif-pan_amd/lib/libddif_old.so"; else lib=""; fi
    python3 bench.py --config wv3_train_b32 --steps 20 --warmup 3 --no-cpu-baseline $lib > gpurun_out/${tag}_$v.json 2> gpurun_out/${tag}_$v.log
    python3 -c "
import json; r=json.load(open('gpurun_out/${tag}_$v.json')); print('$v $rep', 'tiles/s %.1f' % r['value'], 'ms/iter %.2f' % r['ms_per_step'])"
  done
done
