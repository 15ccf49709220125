#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c47
mkdir -p $O
cd $R
L=pcc-rl_amd/lib
timeout 1200 python tools/ab_libraries.py 3 $L/libpcc_sim.so $L/libpcc_sim_ntst.so $L/libpcc_sim_ntld.so $L/libpcc_sim_ntboth.so > $O/ab.txt 2>&1; tail -1 $O/ab.txt
