#!/bin/bash
# round 6, call f: the config-exact goldens (3 distinct T=1000 tiles at B=64, GF2 64x64 DPM-Solver++, CAVE 128x128 T=2000 full chain) + parity reports in both modes
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd $R
(python -m pytest tests/test_gpu_parity.py tests/test_gpu_batch64.py -m gpu -x -q -k "dpm_gf2_64 or full_chain or every_tile_matches" --durations=5 2>&1 | tail -12) > gpurun_out/r06_f_tests.log
cat gpurun_out/r06_f_tests.log
python3 tools/parity_report.py --full > gpurun_out/r06_f_parity_report_f16x2.txt 2>&1; cat gpurun_out/r06_f_parity_report_f16x2.txt
DDIF_F16=0 DDIF_X3=0 python3 tools/parity_report.py --full > gpurun_out/r06_f_parity_report_exact_fp32.txt 2>&1; tail -12 gpurun_out/r06_f_parity_report_exact_fp32.txt
