#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c71
mkdir -p $O
cd $R
PCC_DEBUG_TIMELINE=1 timeout 600 python tools/retire_phases.py 65536 > $O/phases.json 2>$O/err.txt
python - <<'P'
import json,os
d=json.load(open(os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/r06_c71/phases.json'))
for r in d: print(r['step'], {k:round(v,2) for k,v in r['us_per_wavefront'].items()}, {k:(round(v,3) if isinstance(v,float) else v) for k,v in r['search_hints'].items()})
P
