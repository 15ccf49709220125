#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_o
mkdir -p $O
cd $R
for P in 1000 2000; do
PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/exp/libpcc_burn$P.so timeout 300 python tools/sweep3.py '[{}]' > $O/burn$P.log 2>/dev/null; echo burn $P; cat $O/burn$P.log
done
