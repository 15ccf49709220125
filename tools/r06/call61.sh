#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c61
mkdir -p $O
cd $R
PCC_TL_STAGGER=1 PCC_TL_STEPS=150 PCC_DEBUG_TIMELINE=1 timeout 900 python tools/slow_wave_items.py 65536 1 1 3 2>&1 | cut -c1-1200 > $O/stagger.txt; cat $O/stagger.txt
PCC_TL_STEPS=150 PCC_DEBUG_TIMELINE=1 timeout 900 python tools/slow_wave_items.py 65536 1 1 3 2>&1 | cut -c1-1200 > $O/lockstep.txt; cat $O/lockstep.txt
