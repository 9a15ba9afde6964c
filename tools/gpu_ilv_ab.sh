#!/bin/bash
# usage: tools/gpu_ilv_ab.sh <tag>  -- parity tests on the default build, then job time / op timing of the default build (staging
# pieces interleaved into the bf16x3 MFMA loop) against lib/libddif_noilv.so (same sources with -DDDIF_NO_ILV).  Build that one first:
#   cd dif-pan_amd && hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -DDDIF_NO_ILV -c csrc/ddif_plan.cpp -o /tmp/plan_noilv.o
#   hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libddif_noilv.so build/ddif_net.o /tmp/plan_noilv.o build/ddif_lr.o build/ddif_aux.o build/ddif_bwd.o \
#         build/ddif_bwd_ops.o build/ddif_capi.o -Wl,--exclude-libs,ALL
tag=$1
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch64.py -m gpu -q -x -k "not T1000" 2>&1 | tail -6) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
for v in ilv noilv ilv noilv; do
  L=""; [ $v = noilv ] && L="--lib dif-pan_amd/lib/libddif_noilv.so"
  python3 bench.py $L --steps 2 --warmup 1 --T 200 --no-cpu-baseline > $R/gpurun_out/${tag}_bench_$v.json 2> $R/gpurun_out/${tag}_bench_$v.log
  python3 - <<PY
import json
r=json.load(open("$R/gpurun_out/${tag}_bench_$v.json"))
c={x["class"][:12]:round(x["ms_per_step"],3) for x in r["roofline"]["whole_step"]["classes"]}
print("$v ms/denoise-step", round(r["ms_per_step"]/r["config"]["T"],4), c)
PY
done
for v in ilv noilv; do
  L=""; [ $v = noilv ] && L="--lib dif-pan_amd/lib/libddif_noilv.so"
  DDIF_OP_TIMING=$R/gpurun_out/${tag}_op_timing_$v.csv python3 bench.py $L --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
done
