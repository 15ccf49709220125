#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c60
mkdir -p $O
cd $R
PCC_SOAK_BIG=1 timeout 2400 python tests/soak_parity.py 24 2000 0 > $O/soak_big.txt 2>&1; tail -26 $O/soak_big.txt | cut -c1-300
