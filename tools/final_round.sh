#!/bin/bash
# The end-of-round set on one box: the whole GPU suite (timed), smoke, then tools/profile_round.sh TAG.  bash tools/final_round.sh TAG
TAG=${1:-r06_v3}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
( time timeout 1500 python -m pytest tests -m gpu -q ) > $O/gpu_tests.log 2>&1
tail -4 $O/gpu_tests.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
( time timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $O/bench_driver_timed.log 2>&1; grep real $O/bench_driver_timed.log
bash tools/profile_round.sh $TAG > $O/profile_round.log 2>&1
tail -c 400 $O/bench.log
