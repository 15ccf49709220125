#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c42
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q -k "two_sender" > $O/tests.log 2>&1; tail -5 $O/tests.log
PCC_DEBUG_TIMELINE=2 timeout 600 python tools/pass_stats.py '[{}]' 32768 300 2 > $O/pass2.txt 2>&1; tail -1 $O/pass2.txt
timeout 600 python tools/episode_drift.py 32768 2 8 > $O/drift2.txt 2>&1; cat $O/drift2.txt
PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/libpcc_sim_binade.so timeout 600 python tools/episode_drift.py 32768 2 8 > $O/drift2_before.txt 2>&1; cat $O/drift2_before.txt
