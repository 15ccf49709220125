#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c28
mkdir -p $O
cd $R
timeout 900 python tools/ab_step.py '[{"heavy_predict":480},{"heavy_predict":448},{"heavy_predict":416},{"heavy_predict":384},{"heavy_predict":352},{"heavy_predict":320},{"heavy_predict":288},{"heavy_predict":256}]' 65536 4 > $O/ab_hp8.txt 2>&1; cat $O/ab_hp8.txt
timeout 900 python tools/ab_step.py '[{"heavy_predict":384,"heavy_item_packets":1024},{"heavy_predict":384,"heavy_item_packets":1536},{"heavy_predict":384,"heavy_item_packets":2048},{"heavy_predict":384,"send_waves":12},{"heavy_predict":384,"send_waves":14},{"heavy_predict":320,"heavy_item_packets":2048}]' 65536 3 > $O/ab_hp8b.txt 2>&1; cat $O/ab_hp8b.txt
