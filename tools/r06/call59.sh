#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c59
mkdir -p $O
cd $R
timeout 1500 python tests/soak_parity.py 40 1000 0 > $O/soak.txt 2>&1; tail -42 $O/soak.txt | cut -c1-330
