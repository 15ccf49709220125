#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_f
mkdir -p $O
cd $R
timeout 600 python tools/sweep3.py '[{"team_predict":1e18,"heavy_item_packets":0}]' 65536 400 3 > $O/episodes.log 2> $O/episodes.err
cat $O/episodes.log
timeout 1200 python tools/sweep3.py '[{"team_predict":1e18,"heavy_item_packets":0},{"team_predict":1e18,"heavy_item_packets":1024},{"team_predict":1e18,"heavy_item_packets":2048},{"team_predict":4096,"heavy_item_packets":0},{"team_predict":4096,"heavy_item_packets":1024},{"team_predict":6144,"heavy_item_packets":1024},{"team_predict":1e18,"heavy_item_packets":0}]' > $O/sweep.log 2> $O/sweep.err
cat $O/sweep.log
