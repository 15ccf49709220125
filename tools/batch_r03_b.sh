#!/bin/bash
# round-3 batch b: team pass -- parity first, then what it buys
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_b
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "tuning or team or conservation or golden_vectors or philox_batches" > $O/pytest_team.log 2>&1
tail -n 5 $O/pytest_team.log
timeout 600 python tools/sweep2.py '[{"team_predict":1e18},{"team_predict":4096},{"team_predict":2048},{"team_predict":1024},{"team_predict":3072,"heavy_predict":384}]' > $O/sweep.log 2> $O/sweep.err
cat $O/sweep.log
tail -n 3 $O/sweep.err
