#!/bin/bash
# usage: tools/gpu_r3a.sh <tag>  -- round-3 check of the host-side changes: GPU tests, the default bench line (profiled-step accounting),
# the strong-scaling GF2 scene at 64 and 8 tiles per GPU, the self-launcher with --gpus 1 semantics, the training line
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests -m gpu -q -x --durations=15 2>&1 | tail -30) > $R/gpurun_out/${tag}_tests.log 2>&1
cat $R/gpurun_out/${tag}_tests.log
python3 bench.py --steps 3 --warmup 1 --T 200 --no-cpu-baseline > gpurun_out/${tag}_bench_T200.json 2> gpurun_out/${tag}_bench_T200.log
tail -3 gpurun_out/${tag}_bench_T200.log
python3 bench.py --config gf2_dpm50 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_gf2_64.json 2> gpurun_out/${tag}_bench_gf2_64.log
for b in 16 4; do   # squares: what one rank of 4 / 16 would hold ... plus 8 per GPU below through the weak path
  python3 bench.py --config gf2_dpm50 --batch $b --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_gf2_$b.json 2> gpurun_out/${tag}_bench_gf2_$b.log
done
python3 bench.py --config wv3_train_b32 --steps 5 --warmup 2 --cpu-seconds 12 > gpurun_out/${tag}_bench_train.json 2> gpurun_out/${tag}_bench_train.log
tail -2 gpurun_out/${tag}_bench_train.log
python3 - <<PY
import json
for n in ("T200", "gf2_64", "gf2_16", "gf2_4", "train"):
    try:
        r = json.load(open("gpurun_out/${tag}_bench_%s.json" % n))
        rf = r.get("roofline") or {}
        ws = rf.get("whole_step") or {}
        print(n, "value %.5g %s" % (r["value"], r["unit"]), "ms/step %.2f" % r["ms_per_step"], "frac", rf.get("frac"), "classes_sum", ws.get("classes_ms_per_step_sum"), "ms/denoise", ws.get("ms_per_denoising_step"), ws.get("classes_error"))
    except Exception as e:
        print(n, "failed", e)
PY
