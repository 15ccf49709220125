#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c52
mkdir -p $O
cd $R
PCC_TL_STAGGER=1 PCC_TL_STEPS=50,150,250 PCC_DEBUG_TIMELINE=1 timeout 900 python tools/slow_wave_items.py 65536 1 1 10 > $O/slow_stagger.txt 2>&1
python - <<'P'
import json,os
for l in open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06_c52/slow_stagger.txt'):
    if not l.startswith('{'): print(l[:300]); continue
    d=json.loads(l)
    print(d['step'],'span',d['span_us'],'light_last',d['light_last_us'],'wave_items',d['wave_items'],'single',d['single_env_items'],'with_chain',d['single_env_items_with_chain'])
    for r in d['slowest']: print('   ',r['start'],r['finish'],r['envs'],r['packets'],'closed',r['closed'],'sw',r['sweep256'],'chain',r['chain'],'plain',r['plain'],r['ns_per_packet'])
P
