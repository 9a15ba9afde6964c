#!/bin/bash
# round 6, call m: why does ONE ffn.0 instance of four at 64 x 64 take 68-72 us instead of 47?  Arena placement probe: DDIF_ARENA_PAD shifts the relative offsets of the step tensors.
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
for pad in 0 4096 69632 1052672; do
  echo "== DDIF_ARENA_PAD=$pad" >> gpurun_out/r06_m_arena_pad_probe.txt
  DDIF_ARENA_PAD=$pad DDIF_DUMP_PLAN=1 DDIF_OP_TIMING=$R/gpurun_out/r06_m_op_pad$pad.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket 2> gpurun_out/r06_m_plan_pad$pad.log > /dev/null
  grep -E "ffn.0 .*64x64|ffn.3.*64x64" gpurun_out/r06_m_op_pad$pad.csv | cut -d, -f1-3 >> gpurun_out/r06_m_arena_pad_probe.txt
  grep -E "ffn.0 .*@ 64x64" gpurun_out/r06_m_plan_pad$pad.log | sed 's/.*smem=[0-9]*//' | head -4 >> gpurun_out/r06_m_arena_pad_probe.txt
  python3 -c "
import csv; r=list(csv.DictReader(open('gpurun_out/r06_m_op_pad$pad.csv'))); print('sum of op times', round(sum(float(x['us']) for x in r),1), 'us')" >> gpurun_out/r06_m_arena_pad_probe.txt
done
cat gpurun_out/r06_m_arena_pad_probe.txt
