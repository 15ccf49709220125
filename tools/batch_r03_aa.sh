#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_aa
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/pytest.log 2>&1
tail -n 8 $O/pytest.log
timeout 600 python tools/sweep3.py "[{}]" 65536 400 4 auto 2>/dev/null | tee $O/auto.log
