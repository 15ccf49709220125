#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c37
mkdir -p $O
cd $R
PCC_DEBUG_TIMELINE=1 timeout 600 python tools/slow_wave_items.py 32768 2 > $O/slow2.txt 2>&1; cat $O/slow2.txt
