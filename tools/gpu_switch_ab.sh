#!/bin/bash
# usage: tools/gpu_switch_ab.sh <tag> VAR [pytest -k expression]  -- same-box A/B of ONE library switch (VAR=0 against the default) on the default bench at
# T = 200 (twice each, interleaved) + an op-by-op timing of both; preceded by the sampler / forward parity files under the default build
tag=$1; var=$2; kexpr=$3
mkdir -p gpurun_out
if [ -n "$kexpr" ]; then
  (python -m pytest tests/test_gpu_batch64.py tests/test_gpu_parity.py -m gpu -q -k "$kexpr" 2>&1 | tail -12) > gpurun_out/${tag}_tests.log 2>&1
else
  (python -m pytest tests/test_gpu_batch64.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -12) > gpurun_out/${tag}_tests.log 2>&1
fi
cat gpurun_out/${tag}_tests.log
run() {  # name, env...
  name=$1; shift
  env "$@" python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/${tag}_${name}_$rep.json 2> /dev/null
  python3 -c "
import json; r=json.load(open('gpurun_out/${tag}_${name}_$rep.json')); print('$name', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
}
for rep in 1 2; do
  run off $var=0
  run on $var=1
done
DDIF_OP_TIMING=gpurun_out/${tag}_op_timing_on.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
env $var=0 DDIF_OP_TIMING=gpurun_out/${tag}_op_timing_off.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
