#!/bin/bash
# usage: tools/run_pmc_emd.sh <tag> <uniform|step>   (GPU box, repo root): three counter passes over tools/time_emd_one.py
tag=$1; mode=$2
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM GRBM_GUI_ACTIVE" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_WAIT_ANY"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$tag/p$i -- python3 $GRAFT_REPO_ROOT/tools/time_emd_one.py $mode > $GRAFT_REPO_ROOT/gpurun_out/$tag.p$i.log 2>&1
  f=$(find $GRAFT_REPO_ROOT/gpurun_out/$tag/p$i -name "*counter_collection.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $f emd_auction
done
