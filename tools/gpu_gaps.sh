#!/bin/bash
# usage: tools/gpu_gaps.sh <tag> [bench args...]  -- kernel trace of a T=12 job of the default bench; inter-kernel gap statistics of the sampler loop (tools/gaps.py)
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -o p -- python3 $R/bench.py --steps 1 --warmup 1 --T 12 --no-cpu-baseline "$@" > /tmp/tr_$tag.json 2> /tmp/tr_$tag.log
python3 -c "import json; r=json.load(open('/tmp/tr_$tag.json')); print('launches per step', r['config']['launches_per_denoising_step'], 'ms/step (profiled run)', r['roofline']['whole_step']['ms_per_denoising_step'])"
n=$(python3 -c "import json; print(json.load(open('/tmp/tr_$tag.json'))['config']['launches_per_denoising_step'])")
python3 $R/tools/gaps.py $(find /tmp/tr_$tag -name "*kernel_trace.csv") $n | tee $R/gpurun_out/${tag}_gaps.txt
