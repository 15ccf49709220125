#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c69
mkdir -p $O
cd $R
timeout 900 python tools/ab_block.py '[{"retire_wide_predict":256},{"retire_wide_predict":128},{"retire_wide_predict":512},{"retire_wide_predict":1024},{"retire_wide_predict":192}]' 65536 2 1 > $O/ab_rw.txt 2>&1; cat $O/ab_rw.txt
timeout 900 python tools/ab_block.py '[{"retire_grid_frac":0.125},{"retire_grid_frac":0.0625},{"retire_grid_frac":0.25},{"retire_grid_frac":0.5},{"retire_sorted":0}]' 65536 2 1 > $O/ab_rg.txt 2>&1; cat $O/ab_rg.txt
