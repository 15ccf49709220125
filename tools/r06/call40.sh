#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c40
mkdir -p $O
cd $R
timeout 600 python tools/episode_drift.py 32768 2 8 > $O/drift2.txt 2>&1; cat $O/drift2.txt
timeout 600 python tools/episode_drift.py 65536 1 5 > $O/drift1.txt 2>&1; cat $O/drift1.txt
