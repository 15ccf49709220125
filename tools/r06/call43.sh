#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c43
mkdir -p $O
cd $R
timeout 900 python tools/ab_block.py '[{"heavy_predict":640},{"heavy_predict":512},{"heavy_predict":448},{"heavy_predict":384},{"heavy_predict":320}]' 32768 2 2 > $O/ab_hp.txt 2>&1; cat $O/ab_hp.txt
