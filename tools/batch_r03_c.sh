#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_c
mkdir -p $O
cd $R
for T in 1e18 2048; do
PCC_TUNE_TEAM_PREDICT=$T timeout 300 python tools/send_timeline.py > $O/send_tl_$T.json 2> $O/send_tl_$T.err
done
timeout 300 python tools/pass_stats.py '[{"team_predict":1e18},{"team_predict":2048}]' 65536 300 > $O/pass_stats.log 2> $O/pass_stats.err
cat $O/pass_stats.log
tail -n 3 $O/*.err
