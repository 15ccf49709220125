#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c34
mkdir -p $O
cd $R
PCC_DEBUG_TIMELINE=2 timeout 600 python tools/pass_stats.py '[{}]' 32768 300 2 > $O/pass2.txt 2>&1; cat $O/pass2.txt
