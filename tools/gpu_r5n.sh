#!/bin/bash
# round 5, call n: same-box A/B of two builds of the library (default = working tree, other = lib/libddif_old.so) at T = 200, interleaved, three jobs each
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2 3; do
  for v in default other; do
    if [ $v = other ]; then L="--lib $R/dif-pan_amd/lib/libddif_old.so"; else L=""; fi
    python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline $L > gpurun_out/r05_n_${v}_$rep.json 2> /dev/null
    python3 -c "
import json; r=json.load(open('gpurun_out/r05_n_${v}_$rep.json')); print('$v', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
  done
done
