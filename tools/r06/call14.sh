#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c14
mkdir -p $O
cd $R
timeout 600 python tools/ab_step.py '[{"light_wgs":40},{"light_wgs":32},{"light_wgs":24}]' 65536 3 > $O/ab_lw.txt 2>&1; cat $O/ab_lw.txt
timeout 600 python tools/ab_step.py '[{"heavy_predict":384,"heavy_item_packets":1024},{"heavy_predict":448,"heavy_item_packets":1024},{"heavy_predict":512,"heavy_item_packets":1024},{"heavy_predict":384,"heavy_item_packets":1536},{"heavy_predict":320,"heavy_item_packets":1024}]' 65536 3 > $O/ab_hp.txt 2>&1; cat $O/ab_hp.txt
timeout 600 python tools/ab_step.py '[{"send_waves":12},{"send_waves":14},{"send_waves":16}]' 65536 3 > $O/ab_sw.txt 2>&1; cat $O/ab_sw.txt
timeout 600 python tools/ab_step.py '[{"retire_wide_predict":256},{"retire_wide_predict":192},{"retire_wide_predict":384},{"retire_grid_frac":0.125},{"retire_grid_frac":0.25}]' 65536 3 > $O/ab_rt.txt 2>&1; cat $O/ab_rt.txt
