#!/bin/bash
# usage: tools/gpu_pmc_mbench.sh <tag>  -- PMC passes over the microbenchmark's PMC-mode kernels
tag=$1
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VALU SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA"; do
  i=$((i+1)); rm -rf /tmp/pm_$i
  rocprofv3 --pmc $set --output-format csv -d /tmp/pm_$i -o p -- $R/tools/mbench.bin pmc > /tmp/pm_$i.log 2>&1
  python3 $R/tools/pmc_summary.py $(find /tmp/pm_$i -name "*counter_collection.csv") 10 > $R/gpurun_out/${tag}_mb_pmc$i.csv
  cat $R/gpurun_out/${tag}_mb_pmc$i.csv
done
