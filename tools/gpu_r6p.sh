#!/bin/bash
# round 6, call p: four-wave linattn_fused workgroups EVERYWHERE (DDIF_LA_NW=4) against the plan's rule (four only below 256 eight-wave workgroups), B = 64
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for rep in 1 2 3; do
  for v in 0 4; do
    DDIF_LA_NW=$v python3 bench.py --steps 2 --warmup 1 --T 200 --no-cpu-baseline --no-parity --no-bracket 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('LA_NW=$v (0 = the rule) rep $rep ms/step', round(r['ms_per_step']/200,4))" | tee -a gpurun_out/r06_p_la_nw4_all_ab.txt
  done
done
DDIF_LA_NW=4 DDIF_OP_TIMING=$R/gpurun_out/r06_p_op_nw4.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
DDIF_OP_TIMING=$R/gpurun_out/r06_p_op_rule.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
echo "nw4:"; grep linattn_fused gpurun_out/r06_p_op_nw4.csv | cut -d, -f3 | tr '\n' ' '; echo; echo "rule:"; grep linattn_fused gpurun_out/r06_p_op_rule.csv | cut -d, -f3 | tr '\n' ' '; echo
