#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_p
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1
tail -n 12 $O/pytest.log
timeout 900 python tools/sweep3.py '[{},{"retire_wide_predict":0},{"retire_wide_predict":1e18},{"retire_wide_predict":1024},{"retire_wide_predict":256}]' > $O/sweep.log 2> $O/sweep.err
cat $O/sweep.log
