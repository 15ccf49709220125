#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c32
mkdir -p $O
cd $R
timeout 900 python tools/ab_block.py '[{"heavy_predict":640,"send_waves":13},{"heavy_predict":768,"send_waves":13},{"heavy_predict":768,"send_waves":14},{"heavy_predict":896,"send_waves":14},{"heavy_predict":640,"send_waves":14}]' 32768 2 2 > $O/ab_c5.txt 2>&1; cat $O/ab_c5.txt
timeout 900 python tools/ab_block.py '[{"retire_wide_predict":256},{"retire_wide_predict":512},{"retire_wide_predict":1024},{"retire_grid_frac":0.25},{"retire_wide_predict":128}]' 32768 2 2 > $O/ab_c5r.txt 2>&1; cat $O/ab_c5r.txt
