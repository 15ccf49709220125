#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_i
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest_gpu.log 2>&1
tail -n 4 $O/pytest_gpu.log
timeout 900 python tools/sweep3.py '[{},{"group_min_packets":1e18},{"group_min_packets":32},{"group_min_packets":8}]' > $O/sweep.log 2> $O/sweep.err
cat $O/sweep.log
