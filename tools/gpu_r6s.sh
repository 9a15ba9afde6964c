#!/bin/bash
# round 6, call s: the row staging also for the bf16x3 low-resolution 3x3 convs (DDIF_F16=0 inference, every training conv): parity, then the training iteration either way
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_train_graph.py -m gpu -x -q -p no:cacheprovider 2>&1 | tail -3
timeout 900 python3 -m pytest tests/test_env_switches.py -m gpu -x -q -k "F16=0 or X3=0" -p no:cacheprovider 2>&1 | tail -3
for rep in 1 2 3; do
  for v in 0 -; do
    if [ $v = 0 ]; then export DDIF_LR_ROWS=0; else unset DDIF_LR_ROWS; fi
    python3 bench.py --config wv3_train_b32 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('train LR_ROWS=$v rep $rep ms/iteration', round(r['ms_per_step'],3))"
  done
done
