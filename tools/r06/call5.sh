#!/bin/bash
# round 6, GPU call 5: the wave-path threshold re-swept now that the wave path's record stores are coalesced (LDS-staged)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c5
mkdir -p $O
cd $R
timeout 600 python tools/ab_knob.py '[{"heavy_predict":384,"heavy_item_packets":1024},{"heavy_predict":320,"heavy_item_packets":1024},{"heavy_predict":256,"heavy_item_packets":1024},{"heavy_predict":256,"heavy_item_packets":768},{"heavy_predict":192,"heavy_item_packets":768},{"heavy_predict":192,"heavy_item_packets":512},{"heavy_predict":128,"heavy_item_packets":512},{"heavy_predict":512,"heavy_item_packets":1024}]' 65536 4 25 > $O/sweep_hp.txt 2>&1
cat $O/sweep_hp.txt
timeout 600 python tools/ab_knob.py '[{"send_waves":12},{"send_waves":14},{"send_waves":10},{"send_waves":8},{"send_waves":15}]' 65536 3 25 > $O/sweep_sw.txt 2>&1
cat $O/sweep_sw.txt
