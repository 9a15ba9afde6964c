#!/bin/bash
# round 5, call k: XCD-contiguous work partition (DDIF_XCD bit mask: 1 conv, 2 low-resolution, 4 fused linear attention), same-box A/B at T = 200
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2; do
  for v in 0 7 15; do
    DDIF_XCD=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/r05_k_xcd_${v}_$rep.json 2> /dev/null
    python3 -c "
import json; r=json.load(open('gpurun_out/r05_k_xcd_${v}_$rep.json')); print('DDIF_XCD=$v', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
  done
done
