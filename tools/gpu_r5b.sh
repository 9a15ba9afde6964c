#!/bin/bash
# round 5, second call: host-checked mbench r (first block only) + mbench w (sc1 stores / late second-stage loads)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 300 $R/tools/mbench.bin r 2>&1 | grep -v "^  block\|^    [0-9 -]" | head -16 > $R/gpurun_out/r05_b_mbench_r.txt
cat $R/gpurun_out/r05_b_mbench_r.txt
timeout 600 $R/tools/mbench.bin w > $R/gpurun_out/r05_b_mbench_w.txt 2>&1
cat $R/gpurun_out/r05_b_mbench_w.txt
