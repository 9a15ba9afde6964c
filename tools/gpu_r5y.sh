#!/bin/bash
# round 5, call y: the Downsample convs on the f16x2 path (DDIF_S2_F16): forward / sampler parity files, then the default bench at T = 200 with the switch off / on, interleaved
R=$GRAFT_REPO_ROOT
cd $R
python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch64.py tests/test_range_guard.py tests/test_env_switches.py -m gpu -q -x 2>&1 | tail -3
for rep in 1 2 3; do for v in 0 1; do
  DDIF_S2_F16=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.load(sys.stdin); print('DDIF_S2_F16=$v ms per denoising step %.4f' % r['roofline']['whole_step']['ms_per_denoising_step'])"
done; done
DDIF_OP_TIMING=$R/gpurun_out/r05_y_op_timing.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
grep "down\|stem" $R/gpurun_out/r05_y_op_timing.csv
