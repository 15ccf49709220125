#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c64
mkdir -p $O
cd $R
PCC_SOAK_LISTS=1 timeout 1500 python tests/soak_parity.py 40 3000 0 > $O/soak_lists.txt 2>&1; tail -2 $O/soak_lists.txt | cut -c1-300
PCC_SOAK_BIG=1 timeout 1500 python tests/soak_parity.py 12 4000 0 > $O/soak_big.txt 2>&1; tail -2 $O/soak_big.txt | cut -c1-300
