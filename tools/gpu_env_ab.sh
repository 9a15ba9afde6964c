#!/bin/bash
# usage: tools/gpu_env_ab.sh <tag> VAR v1 v2 ...  -- same-box A/B of one library environment switch on the default bench at T = 200 (twice each, interleaved),
# preceded by a parity slice under the LAST value
tag=$1; var=$2; shift; shift
mkdir -p gpurun_out
last=${@: -1}
(env $var=$last python -m pytest tests/test_gpu_batch64.py tests/test_gpu_parity.py -m gpu -q -x -k "forward or ddpm_batch64 or T1000" 2>&1 | tail -3) > gpurun_out/${tag}_tests.log 2>&1
cat gpurun_out/${tag}_tests.log
for rep in 1 2; do
  for v in "$@"; do
    env $var=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/${tag}_${var}_${v}_$rep.json 2> /dev/null
    python3 -c "
import json; r=json.load(open('gpurun_out/${tag}_${var}_${v}_$rep.json')); print('$var=$v', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
  done
done
