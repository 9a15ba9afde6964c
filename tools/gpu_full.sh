#!/bin/bash
# usage: tools/gpu_full.sh <tag> [skip-tests]  -- everything a round's profiles/ needs, in one gpurun call:
#   gpu tests, rocprofv3 kernel stats (T=20), PMC passes (FETCH_SIZE, WRITE_SIZE, MFMA busy; T=4), op-by-op timing (T=40),
#   the default bench (T=1000, with cpu_baseline, the golden parity probe and the exact-fp32 bracket), its bf16 throughput variant, the other BASELINE
#   configurations (gf2_dpm50 with the per-rank shares behind projected_strong_scaling; the training step at 32 and at 4 tiles per GPU), parity reports of both modes
# <tag> = rNN_<series>: the summaries to judge are copied from gpurun_out/<tag>_* to profiles/rNN/<series>_* afterwards (tools/README.md)
tag=$1; shift
R=$GRAFT_REPO_ROOT
rnd=${tag:0:3}; ser=${tag:4}
mkdir -p $R/gpurun_out $R/profiles/$rnd
export DDIF_BUILD_ID=$(python3 -c "import bench; print(bench.build_id())")
if [ "$1" != "skip-tests" ]; then
  (python -m pytest tests -m gpu -q -x 2>&1 | tail -8) > $R/gpurun_out/${tag}_tests.log 2>&1
  cat $R/gpurun_out/${tag}_tests.log
fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$tag -o p -- python3 $R/bench.py --steps 1 --warmup 1 --T 20 --no-cpu-baseline --no-parity --no-bracket > $R/gpurun_out/${tag}_bench_T20.json 2> /tmp/prof_$tag.log
cp $(find /tmp/prof_$tag -name "*kernel_stats.csv") $R/gpurun_out/${tag}_kernel_stats_T20_B64.csv
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python3 $R/bench.py --steps 1 --warmup 0 --T 4 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2> /tmp/pmc_$c.log
  python3 $R/tools/pmc_summary.py $(find /tmp/pmc_$c -name "*counter_collection.csv") 60 > $R/gpurun_out/${tag}_pmc_$c.csv
done
python3 $R/tools/pmc_traffic.py $R/gpurun_out/${tag}_pmc_FETCH_SIZE.csv $R/gpurun_out/${tag}_pmc_WRITE_SIZE.csv $R/gpurun_out/${tag}_hbm_traffic.json
rm -rf /tmp/pmc_mfma
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU --output-format csv -d /tmp/pmc_mfma -o p -- python3 $R/bench.py --steps 1 --warmup 0 --T 4 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2> /tmp/pmc_mfma.log
f=$(find /tmp/pmc_mfma -name "*counter_collection.csv")
if [ -n "$f" ]; then
  python3 $R/tools/pmc_summary.py $f 60 > $R/gpurun_out/${tag}_pmc_mfma.csv
  python3 $R/tools/pmc_mfma.py $f $R/gpurun_out/${tag}_mfma_busy.json
else
  tail -5 /tmp/pmc_mfma.log
fi
cd $R
DDIF_OP_TIMING=$R/gpurun_out/${tag}_op_timing_T40_B64.csv python3 bench.py --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
cp gpurun_out/${tag}_hbm_traffic.json profiles/$rnd/${ser}_hbm_traffic.json 2>/dev/null   # so that the bench line below can tie its traffic to this build
python3 bench.py > gpurun_out/${tag}_bench_T1000_B64.json 2> gpurun_out/${tag}_bench_T1000_B64.log
cat gpurun_out/${tag}_bench_T1000_B64.json
python3 bench.py --config wv3_bf16 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/${tag}_bench_wv3_bf16.json 2> gpurun_out/${tag}_bench_wv3_bf16.log
python3 bench.py --config gf2_dpm50 --steps 3 --warmup 1 --cpu-seconds 12 > gpurun_out/${tag}_bench_gf2_dpm50.json 2> gpurun_out/${tag}_bench_gf2_dpm50.log
# (what one rank of an N-GPU strong-scaling run of the same scene holds -- 32 / 16 / 8 tiles -- is measured by that line itself since round 6: projected_strong_scaling)
python3 bench.py --config cave128_t2000 --steps 1 --warmup 0 --cpu-seconds 12 > gpurun_out/${tag}_bench_cave128_t2000.json 2> gpurun_out/${tag}_bench_cave128_t2000.log
python3 bench.py --config wv3_train_b32 --steps 10 --warmup 3 --cpu-seconds 15 > gpurun_out/${tag}_bench_wv3_train_b32.json 2> gpurun_out/${tag}_bench_wv3_train_b32.log
# BASELINE configs[4] as stated is a GLOBAL batch of 32 over 8 GPUs: the per-rank share is 4 tiles
python3 bench.py --config wv3_train_b32 --batch 4 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_bench_wv3_train_b4_share.json 2> /dev/null
python3 tools/parity_report.py --full > gpurun_out/${tag}_parity_report_f16x2.txt 2>&1
DDIF_F16=0 DDIF_X3=0 python3 tools/parity_report.py --full > gpurun_out/${tag}_parity_report_exact_fp32.txt 2>&1
cd /tmp
rm -rf /tmp/prof_train_$tag
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_train_$tag -o p -- python3 $R/bench.py --config wv3_train_b32 --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2> /tmp/prof_train_$tag.log
cp $(find /tmp/prof_train_$tag -name "*kernel_stats.csv") $R/gpurun_out/${tag}_train_kernel_stats.csv
cd $R
python3 - <<PY
import json
try:
    r = json.load(open("gpurun_out/${tag}_bench_wv3_train_b32.json"))
    print("wv3_train_b32", r["value"], r["unit"], "ms/iteration", r["ms_per_step"])
except Exception as e:
    print("wv3_train_b32 failed", e)
try:
    r = json.load(open("gpurun_out/${tag}_bench_wv3_train_b4_share.json"))
    print("wv3_train per-rank share (4 tiles)", r["value"], r["unit"], "ms/iteration", r["ms_per_step"])
    r = json.load(open("gpurun_out/${tag}_bench_gf2_dpm50.json"))
    print("gf2 shares", r.get("projected_strong_scaling"))
except Exception as e:
    print("shares failed", e)
for n in ("wv3_bf16", "gf2_dpm50", "cave128_t2000"):
    try:
        r = json.load(open("gpurun_out/${tag}_bench_%s.json" % n))
        print(n, r["value"], r["unit"], "ms/job", r["ms_per_step"], "job TF", r["roofline"]["whole_step"]["tflops"])
    except Exception as e:
        print(n, "failed", e)
PY
