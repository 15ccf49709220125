#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c30
mkdir -p $O
cd $R
(timeout 120 tools/microbench/lane_round 0 1024; timeout 120 tools/microbench/lane_round 0 1024 1) > $O/lane_round_wb.txt 2>&1
cat $O/lane_round_wb.txt
