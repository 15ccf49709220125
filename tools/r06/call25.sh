#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c25
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "noise or two_senders" > $O/parity_noise.txt 2>&1; tail -15 $O/parity_noise.txt
timeout 900 python -m pytest tests/test_variants.py -x -q -k "noise" > $O/variants_noise.txt 2>&1; tail -5 $O/variants_noise.txt
timeout 600 python tools/engine_throughput.py 16384 60 > $O/engine_throughput.json 2> $O/et.err; python -c "
import json
d=json.load(open('$O/engine_throughput.json'))
for k,v in d.items():
    if isinstance(v,dict): print(k, v)"
