#!/bin/bash
# round 6, call w: conv_lr_kernel's K loop with the next tap's A fragments prefetched and the products issued product-major over the accumulator blocks:
# bit-identity against the previous build, parity slice, previous build against this one interleaved, per-op table, training iteration
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python3 tools/lib_bits.py run dif-pan_amd/lib/libddif_prev.so gpurun_out/bits_prev.pt 2>&1 | tail -2
python3 tools/lib_bits.py run - gpurun_out/bits_tree.pt 2>&1 | tail -2
python3 tools/lib_bits.py cmp gpurun_out/bits_prev.pt gpurun_out/bits_tree.pt
rm -f gpurun_out/bits_prev.pt gpurun_out/bits_tree.pt
timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_forward_matches_reference_golden or test_ddpm_matches_reference_golden and ddpm_wv3_16_T10 or test_ddim_matches_reference_golden and ddim_gf2 or test_forward_matches_oracle_other_sizes" -p no:cacheprovider 2>&1 | tail -2
DDIF_F16=0 timeout 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_forward_matches_reference_golden" -p no:cacheprovider 2>&1 | tail -2
rm -f gpurun_out/r06_w_lib_ab.txt
bash tools/gpu_lib_ab2.sh r06_w dif-pan_amd/lib/libddif_prev.so 3
for which in other tree; do
  if [ $which = other ]; then L="--lib dif-pan_amd/lib/libddif_prev.so"; else L=""; fi
  DDIF_OP_TIMING=$GRAFT_REPO_ROOT/gpurun_out/r06_w_ops_$which.csv python3 bench.py $L --steps 1 --warmup 1 --T 40 --no-cpu-baseline --no-parity --no-bracket > /dev/null 2>&1
  python3 - <<P
import csv
rows=list(csv.DictReader(open("gpurun_out/r06_w_ops_$which.csv")))
lr=[r for r in rows if r["kernel"].startswith("lr3x3")]; l1=[r for r in rows if r["kernel"].startswith("lr1x1")]
print("$which lr3x3", len(lr), "sum us", round(sum(float(r["us"]) for r in lr),1), "| lr1x1", len(l1), round(sum(float(r["us"]) for r in l1),1), "| all ops us", round(sum(float(r["us"]) for r in rows),1))
P
done
for rep in 1 2; do
  for which in other tree; do
    if [ $which = other ]; then L="--lib dif-pan_amd/lib/libddif_prev.so"; else L=""; fi
    python3 bench.py $L --config wv3_train_b32 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys; r=json.loads(sys.stdin.readline()); print('train $which rep $rep ms/iteration', round(r['ms_per_step'],3))"
  done
done
