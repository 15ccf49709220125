#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c15
mkdir -p $O
cd $R
timeout 900 python tools/ab_step.py '[{"heavy_predict":384,"send_waves":12},{"heavy_predict":448,"send_waves":12},{"heavy_predict":448,"send_waves":14},{"heavy_predict":512,"send_waves":14},{"heavy_predict":576,"send_waves":14},{"heavy_predict":448,"send_waves":15},{"heavy_predict":512,"send_waves":13},{"heavy_predict":448,"send_waves":13}]' 65536 4 > $O/ab_combo.txt 2>&1; cat $O/ab_combo.txt
timeout 900 python tools/ab_step.py '[{"heavy_predict":448,"send_waves":14,"heavy_item_packets":1024},{"heavy_predict":448,"send_waves":14,"heavy_item_packets":768},{"heavy_predict":448,"send_waves":14,"heavy_item_packets":1280},{"heavy_predict":448,"send_waves":14,"team_predict":3072},{"heavy_predict":448,"send_waves":14,"team_predict":6144},{"heavy_predict":480,"send_waves":14}]' 65536 3 > $O/ab_combo2.txt 2>&1; cat $O/ab_combo2.txt
