#!/bin/bash
# round 6, GPU call 1: lane-round microbenchmark, A/B of the send half (round 5's library, the branch-free pipelined lane
# rounds with one / two Philox blocks per loop body), the parity file through the new library, counter attribution.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c1
mkdir -p $O
cd $R
for nb in 0 1 3; do timeout 120 tools/microbench/lane_round $nb; done > $O/lane_round.txt 2>&1
cat $O/lane_round.txt
L=pcc-rl_amd/lib
timeout 900 python tools/ab_libraries.py 3 $L/libpcc_sim_r05.so $L/libpcc_sim.so $L/libpcc_sim_lb2.so > $O/ab_send.txt 2>&1
tail -4 $O/ab_send.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/parity.txt 2>&1
tail -3 $O/parity.txt
timeout 1500 bash tools/r06/pmc_attrib.sh gpurun_out/r06_c1/pmc > $O/pmc.log 2>&1
tail -5 $O/pmc.log
