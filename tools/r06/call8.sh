#!/bin/bash
# round 6, GPU call 8: the whole GPU suite, the driver's bench command, config 5
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c8
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1
tail -5 $O/gpu_tests.txt
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.log 2> $O/bench_driver_style.err
tail -c 3000 $O/bench_driver_style.log
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-policy --no-scaling --config 5 --steps 800 --repeats 1 > $O/bench_config5.log 2>&1
tail -c 1500 $O/bench_config5.log
