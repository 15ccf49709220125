TAG=r05_v3
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
timeout 1000 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1
tail -3 $O/gpu_tests.log
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd /tmp && export TMPDIR=/tmp
timeout 400 python $R/bench.py > $O/bench.log 2> $O/bench.err
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o st -- python $R/bench.py --no-cpu-baseline --no-pmc --no-policy > $O/stats.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 400 --warmup 20 --repeats 1 --no-cpu-baseline --no-pmc --no-policy > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 400 --warmup 20 --repeats 1 --no-cpu-baseline --no-pmc --no-policy > $O/pmc_write.log 2>&1
python $R/tools/pmc_aggregate.py $O/pmc_hbm.json $O/pmc_fetch $O/pmc_write > /dev/null
cd $R
timeout 200 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_style.log 2>&1
timeout 200 python $R/bench.py --no-cpu-baseline --no-pmc --no-policy --config 5 --steps 800 --repeats 1 > $O/bench_config5.log 2>&1
timeout 200 python $R/bench.py --no-cpu-baseline --no-pmc --no-policy --stagger --steps 800 --repeats 1 > $O/bench_stagger.log 2>&1
find $O -name "*kernel_stats.csv" | head -3
tail -c 1500 $O/bench.log
