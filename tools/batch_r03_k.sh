#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_k
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tuning or team or conservation or philox" > $O/pytest.log 2>&1
tail -n 3 $O/pytest.log
timeout 1500 python tools/sweep3.py '[{"group_min_packets":1e18},{"group_min_packets":200},{"group_min_packets":300},{"group_min_packets":1e18,"heavy_predict":384},{"group_min_packets":1e18,"heavy_predict":320},{"group_min_packets":1e18,"heavy_predict":256},{"group_min_packets":1e18,"heavy_predict":256,"heavy_item_packets":1024},{"group_min_packets":1e18,"heavy_predict":384,"heavy_item_packets":4096},{"group_min_packets":1e18,"team_predict":1e18}]' > $O/sweep.log 2> $O/sweep.err
cat $O/sweep.log
