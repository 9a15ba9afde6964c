#!/bin/bash
# round 6, call q: 8 x 16 tiles EVERYWHERE (DDIF_TILE16=0) against the plan's rule at B = 64 (quick check: +0.6 %, the rule stays)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for v in - 0; do
    if [ $v = 0 ]; then export DDIF_TILE16=0; else unset DDIF_TILE16; fi
    python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('TILE16=$v rep $rep ms/step', round(r['ms_per_step']/200,4))"
  done
done
