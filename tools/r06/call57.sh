#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c57
mkdir -p $O
cd $R
hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/microbench/store_bw.hip -o /tmp/store_bw 2>/dev/null
for w in 1 4 8; do timeout 120 /tmp/store_bw $w 1536; done > $O/store_bw.txt 2>&1
timeout 120 /tmp/store_bw 8 256 >> $O/store_bw.txt 2>&1
cat $O/store_bw.txt
