#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c68
mkdir -p $O
cd $R
L=pcc-rl_amd/lib
timeout 1200 python tools/ab_libraries.py 4 $L/libpcc_sim_head.so $L/libpcc_sim.so > $O/ab1.txt 2>&1; tail -1 $O/ab1.txt
AB_ENVS=32768 AB_SENDERS=2 timeout 1200 python tools/ab_libraries.py 3 $L/libpcc_sim_head.so $L/libpcc_sim.so > $O/ab2.txt 2>&1; tail -1 $O/ab2.txt
