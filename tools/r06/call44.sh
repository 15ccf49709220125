#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c44
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_full_size.py -m gpu -x -q -k "two_sender" > $O/tests.log 2>&1; tail -3 $O/tests.log
AB_ENVS=32768 AB_SENDERS=2 timeout 600 python tools/ab_libraries.py 4 pcc-rl_amd/lib/libpcc_sim_head.so pcc-rl_amd/lib/libpcc_sim.so > $O/ab.txt 2>&1; tail -1 $O/ab.txt
PCC_TL_STEPS=3,8,15,25 PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY= timeout 600 python tools/slow_wave_items.py 32768 2 1 6 > $O/slow_early.txt 2>&1
python - <<'P'
import json,os
for l in open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06_c44/slow_early.txt'):
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(d['step'],d['span_us'],d['light_last_us'],d['wave_items'],d['single_env_items_with_chain'],{k:v for k,v in d['refusals_over_items_with_chain'].items() if v}, [(r['finish'],r['packets'],r['closed'],r['sweep256'],r['chain'],r['plain'],r['refused_by']) for r in d['slowest']])
P
