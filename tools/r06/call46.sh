#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c46
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py tests/test_variants.py -m gpu -x -q -k "two_sender or config5 or senders" > $O/tests.log 2>&1; tail -3 $O/tests.log
timeout 600 python tools/step_times.py 32768 2 4 > $O/steps2.txt 2>&1; python - <<'P'
import json,os
for l in open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06_c46/steps2.txt'):
    if l.startswith('{'):
        d=json.loads(l); print(d['launch'],d['mean'],d['p10_p50_p90_p99_max'],d['mean_by_stretch_of_the_episode'])
P
AB_ENVS=32768 AB_SENDERS=2 timeout 600 python tools/ab_libraries.py 3 pcc-rl_amd/lib/libpcc_sim_head.so pcc-rl_amd/lib/libpcc_sim.so > $O/ab.txt 2>&1; tail -1 $O/ab.txt
PCC_DEBUG_TIMELINE=2 timeout 600 python tools/pass_stats.py '[{}]' 32768 300 2 > $O/pass2.txt 2>&1; tail -1 $O/pass2.txt
