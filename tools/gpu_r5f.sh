#!/bin/bash
# round 5, call f: RCCL world-size-1 tests + same-box A/B of the library with the range watch against the build before it (lib/libddif_old.so)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
(python -m pytest tests/test_rccl_world1.py -m gpu -q -x 2>&1 | tail -25) > $R/gpurun_out/r05_f_rccl_tests.log 2>&1
cat $R/gpurun_out/r05_f_rccl_tests.log
if [ -f dif-pan_amd/lib/libddif_old.so ]; then bash tools/gpu_lib_ab.sh r05_f dif-pan_amd/lib/libddif_old.so; fi
