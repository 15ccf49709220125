#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c29
mkdir -p $O
cd $R
timeout 900 python tools/ab_block.py '[{"heavy_predict":480},{"heavy_predict":384},{"heavy_predict":320},{"heavy_predict":600},{"heavy_predict":256}]' 65536 2 > $O/ab_pos8.txt 2>&1; cat $O/ab_pos8.txt
PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/libpcc_sim_pos4.so timeout 900 python tools/ab_block.py '[{"heavy_predict":480},{"heavy_predict":384},{"heavy_predict":320},{"heavy_predict":600},{"heavy_predict":256}]' 65536 2 > $O/ab_pos4.txt 2>&1; cat $O/ab_pos4.txt
