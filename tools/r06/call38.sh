#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c38
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q -k "two_sender" > $O/tests.log 2>&1; tail -5 $O/tests.log
PCC_DEBUG_TIMELINE=1 timeout 600 python tools/slow_wave_items.py 32768 2 2>&1 | cut -c1-1500 > $O/slow2.txt; cat $O/slow2.txt
AB_ENVS=32768 AB_SENDERS=2 timeout 600 python tools/ab_libraries.py 3 pcc-rl_amd/lib/libpcc_sim_norelax.so pcc-rl_amd/lib/libpcc_sim.so > $O/ab.txt 2>&1; tail -2 $O/ab.txt
