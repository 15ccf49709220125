#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_g
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -x -q -m gpu > $O/pytest_gpu.log 2>&1
tail -n 6 $O/pytest_gpu.log
timeout 900 python tools/sweep3.py '[{},{"team_predict":1e18,"heavy_item_packets":0},{"heavy_item_packets":1024},{"heavy_item_packets":3072}]' > $O/sweep.log 2> $O/sweep.err
cat $O/sweep.log
