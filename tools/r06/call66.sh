#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c66
mkdir -p $O
cd $R
PCC_DEBUG_TIMELINE=1 timeout 600 python tools/retire_timeline.py 65536 > $O/retire_tl.json 2>$O/err.txt
python - <<'P'
import json,os
d=json.load(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06_c66/retire_tl.json'))
for r in (d if isinstance(d,list) else d.get('steps',[])):
    print(r['step'],'span',round(r['span_us'],1),'wgs',r['workgroups'],'mean_running',round(r['mean_running']),'dur',{k:round(v,1) for k,v in r['dur_us'].items()},'last_start',round(r['last_start_us'],1),'start p50/p90/p99',[round(x,1) for x in r['start_us_p50_p90_p99']])
    print('   running',r['running_over_time'])
    print('   last',[(x['wg'],round(x['start'],1),round(x['dur'],1)) for x in r['last_to_finish']])
P
