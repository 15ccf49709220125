#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c45
mkdir -p $O
cd $R
timeout 600 python tools/step_times.py 65536 1 4 > $O/steps1.txt 2>&1; cat $O/steps1.txt
timeout 600 python tools/step_times.py 32768 2 4 > $O/steps2.txt 2>&1; cat $O/steps2.txt
