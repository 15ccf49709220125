#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c51
mkdir -p $O
cd $R
L=pcc-rl_amd/lib
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_variants.py -m gpu -x -q -k "two_sender or config5 or senders or golden or full_size" > $O/tests.log 2>&1; tail -3 $O/tests.log
AB_ENVS=32768 AB_SENDERS=2 timeout 1200 python tools/ab_libraries.py 3 $L/libpcc_sim_head.so $L/libpcc_sim.so > $O/ab2.txt 2>&1; tail -1 $O/ab2.txt
timeout 1200 python tools/ab_libraries.py 3 $L/libpcc_sim_head.so $L/libpcc_sim.so > $O/ab1.txt 2>&1; tail -1 $O/ab1.txt
