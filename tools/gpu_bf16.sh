#!/bin/bash
# usage: tools/gpu_bf16.sh <tag>  -- the bf16 throughput variant on one box: its tests, `bench.py --config wv3_bf16` (full T = 1000, drift included) next to the
# default configuration at the same T, and an op-by-op timing of a bf16 step
tag=$1
mkdir -p gpurun_out
(python -m pytest tests/test_bf16_variant.py -m gpu -q -s 2>&1 | tail -15) > gpurun_out/${tag}_tests.log 2>&1
cat gpurun_out/${tag}_tests.log
python3 bench.py --config wv3_bf16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_wv3_bf16.json 2> gpurun_out/${tag}_bench_wv3_bf16.log
tail -4 gpurun_out/${tag}_bench_wv3_bf16.log
python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_wv3_same_box.json 2> gpurun_out/${tag}_bench_wv3_same_box.log
python3 - <<P
import json
for n in ("wv3_bf16", "wv3_same_box"):
    r = json.load(open("gpurun_out/${tag}_bench_%s.json" % n))
    ws = r["roofline"]["whole_step"]
    print(n, "MP/s %.4f" % r["value"], "ms/denoise %.3f" % ws["ms_per_denoising_step"], r["dtype"], r["roofline"]["bound"], "frac %.3f" % r["roofline"]["frac"],
          [(c["class"][:12], round(c["ms_per_step"], 3)) for c in (ws.get("classes") or [])], r.get("drift"))
P
DDIF_MATH=bf16 DDIF_OP_TIMING=gpurun_out/${tag}_op_timing_bf16.csv python3 bench.py --config wv3_bf16 --steps 1 --warmup 1 --T 40 --no-cpu-baseline > /dev/null 2>&1
