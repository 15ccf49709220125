#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_j
mkdir -p $O
cd $R
for T in 16 1e18; do
PCC_TUNE_GROUP_MIN_PACKETS=$T timeout 300 python tools/send_timeline.py > $O/send_tl_$T.json 2> $O/send_tl_$T.err
done
timeout 300 python tools/pass_stats.py '[{"group_min_packets":16},{"group_min_packets":1e18}]' 65536 100 > $O/pass_stats.log 2> $O/pass_stats.err
cat $O/pass_stats.log
for S in 0 4 8 12; do
PCC_DEBUG_SKIP=$S timeout 200 python bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip $S', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['other_kernels'][0]['kernel_ms'])"
done
