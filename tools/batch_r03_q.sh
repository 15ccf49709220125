#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_q
mkdir -p $O
cd $R
timeout 2000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1
tail -n 5 $O/pytest.log
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_short.log 2> $O/bench_short.err; tail -n 1 $O/bench_short.log | cut -c 1-1800
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --config 2 > $O/bench_cfg2.log 2>> $O/bench_short.err; tail -n 1 $O/bench_cfg2.log | cut -c 1-700
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --config 5 > $O/bench_cfg5.log 2>> $O/bench_short.err; tail -n 1 $O/bench_cfg5.log | cut -c 1-700
tail -n 5 $O/bench_short.err
