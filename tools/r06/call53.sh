#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c53
mkdir -p $O
cd $R
timeout 900 python tools/ab_block.py '[{"send_envs_per_wave":64},{"send_envs_per_wave":48},{"send_envs_per_wave":32},{"send_envs_per_wave":40},{"send_envs_per_wave":56}]' 65536 2 1 > $O/ab_epw.txt 2>&1; cat $O/ab_epw.txt
