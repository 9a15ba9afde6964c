#!/bin/bash
# round 5, call t: weight-gradient microbenchmark (reference = the batch-loading fp32 kernel) + training bench against the number of workgroups a wgrad launch aims for
R=$GRAFT_REPO_ROOT
cd $R
timeout 200 tools/mbench_wgrad.bin | grep "max |dW"
for rep in 1 2; do for w in 512 384 256; do
  DDIF_WGRAD_WGS=$w python3 bench.py --config wv3_train_b32 --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; r=json.load(sys.stdin); print('DDIF_WGRAD_WGS=$w training ms/iteration %.2f' % r['ms_per_step'])"
done; done
