#!/bin/bash
# round 5, call p: the whole round on ONE box -- round 4's tree (copied to _r04/, its own bench.py + library) against this tree, default bench at T = 200, interleaved
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
for rep in 1 2 3; do
  (cd $R/_r04 && python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline 2>/dev/null) > $R/gpurun_out/r05_p_r04_$rep.json
  python3 -c "
import json; r=json.load(open('$R/gpurun_out/r05_p_r04_$rep.json')); print('round 4 tree', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
  (cd $R && python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline 2>/dev/null) > $R/gpurun_out/r05_p_r05_$rep.json
  python3 -c "
import json; r=json.load(open('$R/gpurun_out/r05_p_r05_$rep.json')); print('round 5 tree', $rep, 'ms/denoise %.3f' % r['roofline']['whole_step']['ms_per_denoising_step'], [(c['class'][:12], round(c['ms_per_step'],3)) for c in (r['roofline']['whole_step']['classes'] or [])])"
done
(cd $R/_r04 && python3 bench.py --config gf2_dpm50 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null) | python3 -c "import json,sys; r=json.load(sys.stdin); print('round 4 gf2 8 tiles ms/job', r['ms_per_step'])"
(cd $R && python3 bench.py --config gf2_dpm50 --batch 8 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null) | python3 -c "import json,sys; r=json.load(sys.stdin); print('round 5 gf2 8 tiles ms/job', r['ms_per_step'])"
(cd $R/_r04 && python3 bench.py --config wv3_train_b32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null) | python3 -c "import json,sys; r=json.load(sys.stdin); print('round 4 training ms/iteration', r['ms_per_step'])"
(cd $R && python3 bench.py --config wv3_train_b32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null) | python3 -c "import json,sys; r=json.load(sys.stdin); print('round 5 training ms/iteration', r['ms_per_step'])"
