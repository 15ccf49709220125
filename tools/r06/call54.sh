#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c54
mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I include -I pcc-rl_amd/csrc tools/microbench/lane_round.hip -o /tmp/lane_round 2>/dev/null
for lanes in 64 32 16 8; do echo "== one wavefront per compute unit, $lanes lanes with packets"; timeout 300 /tmp/lane_round 0 256 0 $lanes | grep -E "variant (0|2|5) "; done > $O/lanes.txt 2>&1
for lanes in 64 32; do echo "== one wavefront per SIMD (workgroups of 4), $lanes lanes with packets"; timeout 300 /tmp/lane_round 0 1024 0 $lanes | grep -E "variant (2|5) "; done >> $O/lanes.txt 2>&1
cat $O/lanes.txt
