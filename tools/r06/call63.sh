#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c63
mkdir -p $O
cd $R
timeout 900 python tools/ab_block.py '[{"light_front":0},{"light_front":2},{"light_front":4},{"light_front":6},{"light_front":10}]' 65536 3 1 > $O/ab_lf.txt 2>&1; cat $O/ab_lf.txt
timeout 900 python tools/ab_block.py '[{"light_front":0},{"light_front":2},{"light_front":4},{"light_front":8}]' 32768 2 2 > $O/ab_lf2.txt 2>&1; cat $O/ab_lf2.txt
