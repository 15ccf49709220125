#!/bin/bash
# The round's measurement set, run on the GPU box from the repo root:  bash tools/profile_round.sh TAG
# bench line (with the CPU baselines), rocprofv3 kernel stats of the same command, HBM PMC passes,
# send critical path, retire phases.  Results under gpurun_out/TAG/ (copy what is to be kept into profiles/).
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export PCC_COMMIT=$(python -c "import sys; sys.path.insert(0, '$R'); import bench; print(bench.code_stamp())" 2>/dev/null)
timeout 600 python $R/bench.py > $O/bench.log 2> $O/bench.err
timeout 600 python $R/bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.log 2> $O/bench_driver_style.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python $R/bench.py --no-cpu-baseline --no-pmc --no-policy --no-scaling > $O/stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 400 --warmup 20 --repeats 1 --no-cpu-baseline --no-pmc --no-policy --no-scaling > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 400 --warmup 20 --repeats 1 --no-cpu-baseline --no-pmc --no-policy --no-scaling > $O/pmc_write.log 2>&1
python $R/tools/pmc_aggregate.py $O/pmc_hbm.json $O/pmc_fetch $O/pmc_write > /dev/null
timeout 200 python $R/bench.py --no-cpu-baseline --no-pmc --no-policy --no-scaling --stagger --steps 800 --repeats 1 > $O/bench_stagger.log 2>&1
cd $R
timeout 200 python $R/bench.py --no-cpu-baseline --no-pmc --no-policy --no-scaling --config 5 --steps 800 --repeats 1 > $O/bench_config5.log 2>&1
timeout 200 python $R/bench.py --no-cpu-baseline --no-pmc --no-policy --no-scaling --config 2 --repeats 1 > $O/bench_config2.log 2>&1
timeout 200 python tools/send_timeline.py > $O/send_critical_path.json 2> $O/tl.err
timeout 300 python tools/engine_throughput.py 16384 60 > $O/engine_throughput.json 2>> $O/tl.err
timeout 300 python tools/ppo_throughput.py > $O/ppo_throughput.json 2>> $O/tl.err
find $O -name "*kernel_stats.csv" | head -3
tail -c 2500 $O/bench.log
