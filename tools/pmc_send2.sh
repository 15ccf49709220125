#!/bin/bash
# two SQ passes only: tools/pmc_send2.sh OUTDIR case waves steps
O=$1; CASE=$2; WAVES=$3; STEPS=${4:-50}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_WR" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_WAVES SQ_IFETCH SQ_INSTS_LDS SQ_INSTS_VALU_INT64 SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/tools/light_only.py $CASE $WAVES $STEPS > $O/p$i.log 2>&1
done
