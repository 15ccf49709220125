#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c4
mkdir -p $O
cd $R
(for np in 32 256 1024; do timeout 120 tools/microbench/lane_round 0 $np; done; timeout 120 tools/microbench/lane_round 1 1024) > $O/lane_round_t.txt 2>&1
cat $O/lane_round_t.txt
