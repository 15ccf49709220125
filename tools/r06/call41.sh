#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c41
mkdir -p $O
cd $R
PCC_DEBUG_TIMELINE=1 timeout 900 python tools/slow_wave_items.py 32768 2 8 3 > $O/slow2.txt 2>&1
python - <<'P'
import json,os
for l in open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06_c41/slow2.txt'):
    if not l.startswith('{'): continue
    d=json.loads(l)
    print(d['episode'],d['step'],d['span_us'],d['light_last_us'],d['single_env_items_with_chain'],{k:v for k,v in d['refusals_over_items_with_chain'].items() if v}, [(r['finish'],r['packets'],r['closed'],r['chain'],r['plain'],r['refused_by']) for r in d['slowest']])
P
