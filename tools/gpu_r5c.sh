#!/bin/bash
# round 5, call c: the resident ResnetBlock kernel of the 8 x 8 level against the two conv_lr launches it replaces (timing + host check)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
timeout 600 $R/tools/rb/mbench_rb.bin > $R/gpurun_out/r05_c_mbench_rb.txt 2>&1
cat $R/gpurun_out/r05_c_mbench_rb.txt
