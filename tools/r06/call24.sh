#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R
for v in 0 1 2 3; do
  L=$R/pcc-rl_amd/lib/libpcc_sim_ns2v$v.so; [ $v = 0 ] && L=$R/pcc-rl_amd/lib/libpcc_sim.so
  echo "variant $v: $(PCC_SIM_LIBRARY=$L timeout 200 python tools/noise_profile.py 16384 60 2 2>&1 | grep 'ms per step')"
done
