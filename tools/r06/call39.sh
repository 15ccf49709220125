#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c39
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "powers_of_two" > $O/tests.log 2>&1; tail -3 $O/tests.log
PCC_DEBUG_TIMELINE=1 timeout 600 python tools/slow_wave_items.py 65536 1 2>&1 | cut -c1-2500 > $O/slow1.txt; cat $O/slow1.txt
PCC_DEBUG_TIMELINE=2 timeout 600 python tools/pass_stats.py '[{}]' 65536 300 1 > $O/pass1.txt 2>&1; tail -1 $O/pass1.txt
timeout 300 python bench.py --config 5 --no-scaling > $O/bench_c5.log 2>&1; tail -1 $O/bench_c5.log | cut -c1-300
