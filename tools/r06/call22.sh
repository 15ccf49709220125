#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c22
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "noise or two_senders" > $O/parity_noise.txt 2>&1; tail -30 $O/parity_noise.txt
timeout 600 python tools/engine_throughput.py 16384 60 > $O/engine_throughput.json 2> $O/et.err; cat $O/engine_throughput.json | head -60
