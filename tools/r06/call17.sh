#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c17
mkdir -p $O
cd $R
timeout 900 python tools/ab_step.py '[{"heavy_predict":640,"send_waves":13},{"heavy_predict":640,"send_waves":12},{"heavy_predict":768,"send_waves":13},{"heavy_predict":896,"send_waves":13},{"heavy_predict":1024,"send_waves":13},{"heavy_predict":768,"send_waves":14},{"heavy_predict":512,"send_waves":13}]' 32768 4 2 > $O/ab_c5.txt 2>&1; cat $O/ab_c5.txt
timeout 900 python tools/ab_step.py '[{"heavy_item_packets":1024},{"heavy_item_packets":768},{"heavy_item_packets":1536},{"heavy_item_packets":2048},{"retire_wide_predict":256},{"retire_wide_predict":512},{"retire_wide_predict":128}]' 32768 3 2 > $O/ab_c5b.txt 2>&1; cat $O/ab_c5b.txt
