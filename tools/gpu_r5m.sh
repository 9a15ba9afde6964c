#!/bin/bash
# round 5, call m: zig-zag traversal (DDIF_ZIGZAG=0 / 1), same-box A/B at T = 200 + bit-identity of the results
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2; do
  for v in 0 1; do
    DDIF_ZIGZAG=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/r05_m_zz_${v}_$rep.json 2> /dev/null
    python3 -c "
import json; r=json.load(open('gpurun_out/r05_m_zz_${v}_$rep.json')); print('DDIF_ZIGZAG=$v', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
  done
done
(python -m pytest tests/test_env_switches.py -m gpu -q -x -k "resident_weights or bit_identical" 2>&1 | tail -5)
