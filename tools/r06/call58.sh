#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c58
mkdir -p $O
cd $R
timeout 900 python tools/ab_block.py '[{"heavy_predict":480},{"heavy_predict":400},{"heavy_predict":320},{"heavy_predict":256},{"heavy_predict":560}]' 65536 2 1 > $O/ab_hp.txt 2>&1; cat $O/ab_hp.txt
timeout 900 python tools/ab_block.py '[{"send_waves":13},{"send_waves":12},{"send_waves":14},{"send_waves":15},{"send_waves":16}]' 65536 2 1 > $O/ab_sw.txt 2>&1; cat $O/ab_sw.txt
