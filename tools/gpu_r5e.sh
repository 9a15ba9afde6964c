#!/bin/bash
# round 5, call e: range-guard tests on the MI355X + the default bench at T = 200 (cost of the range watch: compare with r05_a's 3.89 ms on HIP_FORCE_DEV_KERNARG=1)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests/test_range_guard.py -m gpu -q -x 2>&1 | tail -15) > $R/gpurun_out/r05_e_range_tests.log 2>&1
cat $R/gpurun_out/r05_e_range_tests.log
for rep in 1 2; do
python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/r05_e_bench_$rep.json 2> /dev/null
python3 -c "
import json; r=json.load(open('gpurun_out/r05_e_bench_$rep.json')); print('ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], 'step_frac', r['roofline'].get('step_frac'), 'floor', r['roofline'].get('step_floor_ms'), r['dtype'])"
done
