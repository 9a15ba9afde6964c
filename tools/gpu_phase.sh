#!/bin/bash
# usage: tools/gpu_phase.sh <tag> -- phase times of a training iteration + the train tests + the config-5 bench line
tag=$1; shift
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
python3 tools/train_phase_times.py 32 > gpurun_out/${tag}_phase_times.txt 2>&1
cat gpurun_out/${tag}_phase_times.txt
bash tools/gpu_train.sh $tag
