#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c70
mkdir -p $O
cd $R
L=pcc-rl_amd/lib
timeout 1200 python tools/ab_libraries.py 3 $L/libpcc_sim.so $L/libpcc_sim_ilp.so $L/libpcc_sim_memclause.so > $O/ab1.txt 2>&1; tail -1 $O/ab1.txt
