#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_m
mkdir -p $O
cd $R
timeout 300 python tools/sweep3.py '[{}]' > $O/base.log 2>/dev/null; cat $O/base.log
for P in 1 3; do
PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/exp/libpcc_prio$P.so timeout 300 python tools/sweep3.py '[{}]' > $O/prio$P.log 2>/dev/null; echo prio $P; cat $O/prio$P.log
done
