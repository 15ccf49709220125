#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c49
mkdir -p $O
cd $R
L=pcc-rl_amd/lib
timeout 1200 python tools/ab_libraries.py 4 $L/libpcc_sim.so $L/libpcc_sim_ntrun.so $L/libpcc_sim_ntrunld.so > $O/ab.txt 2>&1; cat $O/ab.txt | tail -13
