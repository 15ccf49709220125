#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_ae
mkdir -p $O
cd $R
timeout 2000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -n 6 $O/pytest.log
for C in 2 5 3; do
timeout 300 python bench.py --steps 400 --warmup 50 --repeats 2 --no-cpu-baseline --config $C 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('config $C value %.3e'%d['value'], 'ms', round(d['ms_per_step'],4), 'send', round(r['kernel_ms'],4), 'retire', round(r['other_kernels'][0]['kernel_ms'],4), 'pk', round(d['config']['packets_per_env_step'],1))"
done
