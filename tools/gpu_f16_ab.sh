#!/bin/bash
# usage: tools/gpu_f16_ab.sh <tag>  -- same-box A/B of the split-operand paths on the default bench at T = 200 (twice each, interleaved):
# bf16x3 (DDIF_F16=0) against f16x2 (default); preceded by the sampler / forward parity files under the default
tag=$1
mkdir -p gpurun_out
(python -m pytest tests/test_gpu_batch64.py tests/test_gpu_parity.py -m gpu -q 2>&1 | tail -12) > gpurun_out/${tag}_tests.log 2>&1
cat gpurun_out/${tag}_tests.log
run() {  # name, env...
  name=$1; shift
  env "$@" python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/${tag}_${name}_$rep.json 2> /dev/null
  python3 -c "
import json; r=json.load(open('gpurun_out/${tag}_${name}_$rep.json')); print('$name', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
}
for rep in 1 2; do
  run x3 DDIF_F16=0
  run f16 DDIF_F16=1
done
DDIF_OP_TIMING=gpurun_out/${tag}_op_timing_f16.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
DDIF_F16=0 DDIF_OP_TIMING=gpurun_out/${tag}_op_timing_x3.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
