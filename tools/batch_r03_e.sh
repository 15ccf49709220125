#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_e
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tuning or team or conservation" > $O/pytest_team.log 2>&1
tail -n 3 $O/pytest_team.log
timeout 900 python tools/sweep2.py '[{"team_predict":1e18,"heavy_item_packets":0},{"team_predict":1e18,"heavy_item_packets":1024},{"team_predict":1e18,"heavy_item_packets":2048},{"team_predict":1e18,"heavy_item_packets":4096},{"team_predict":4096,"heavy_item_packets":2048},{"team_predict":6144,"heavy_item_packets":2048},{"team_predict":3072,"heavy_item_packets":2048},{"team_predict":4096,"heavy_item_packets":2048,"heavy_predict":384}]' > $O/sweep.log 2> $O/sweep.err
cat $O/sweep.log
