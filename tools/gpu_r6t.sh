#!/bin/bash
# round 6, call t: the step counter's wait moved behind the load burst (dd_late) + scalar copies of c0 / c1 in the low-resolution kernel (no vector load from the
# kernel-argument segment in front of the staging loads): parity slice, then the previous build against this one, interleaved; then the training iteration likewise
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_forward_matches_reference_golden or test_ddpm_matches_reference_golden and ddpm_wv3_16_T10 or test_ddim_matches_reference_golden and ddim_gf2 or test_forward_matches_oracle_other_sizes" -p no:cacheprovider 2>&1 | tail -3
DDIF_F16=0 timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_forward_matches_reference_golden" -p no:cacheprovider 2>&1 | tail -3
rm -f gpurun_out/r06_t_lib_ab.txt
bash tools/gpu_lib_ab2.sh r06_t dif-pan_amd/lib/libddif_prev.so 3
for rep in 1 2; do
  for which in other tree; do
    if [ $which = other ]; then L="--lib dif-pan_amd/lib/libddif_prev.so"; else L=""; fi
    python3 bench.py $L --config wv3_train_b32 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('train $which rep $rep ms/iteration', round(r['ms_per_step'],3))"
  done
done
