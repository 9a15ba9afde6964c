#!/bin/bash
# round 5, call g: resident-weights tiling in the plan: bit-identity test + same-box A/B (DDIF_WRES=0 / 1) at T = 200
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests/test_env_switches.py -m gpu -q -x -k "resident_weights" 2>&1 | tail -15) > $R/gpurun_out/r05_g_wres_test.log 2>&1
cat $R/gpurun_out/r05_g_wres_test.log
for rep in 1 2; do
  for v in 0 1; do
    DDIF_WRES=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/r05_g_wres_${v}_$rep.json 2> /dev/null
    python3 -c "
import json; r=json.load(open('gpurun_out/r05_g_wres_${v}_$rep.json')); print('DDIF_WRES=$v', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
  done
done
DDIF_WRES=1 python3 bench.py --config gf2_dpm50 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.load(sys.stdin); print('gf2 8 tiles ms/job', r['ms_per_step'])"
