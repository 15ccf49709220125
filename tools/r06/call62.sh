#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c62
mkdir -p $O
cd $R
timeout 900 python tools/ab_block.py '[{"light_front":0},{"light_front":4},{"light_front":8},{"light_front":16},{"light_front":32}]' 65536 2 1 > $O/ab_lf.txt 2>&1; cat $O/ab_lf.txt
for lf in 0 8 32 0 8 32; do PCC_BENCH_TUNING="{\"light_front\": $lf}" timeout 300 python bench.py --stagger --steps 800 --repeats 1 --no-cpu-baseline --no-pmc --no-policy --no-scaling 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']; print('stagger light_front $lf', round(d['ms_per_step'],4), 'send', round(r['kernel_ms'],4), 'retire', round(r['other_kernels'][0]['kernel_ms'],4))"; done > $O/stagger.txt 2>&1; cat $O/stagger.txt
