#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_n
mkdir -p $O
cd $R
for N in 16384 32768 131072; do
timeout 300 python tools/sweep3.py '[{}]' $N > $O/n$N.log 2>/dev/null; echo N $N; cat $O/n$N.log
done
cd /tmp && export TMPDIR=/tmp
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o f -- python $R/bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline > $O/pmc_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o w -- python $R/bench.py --steps 100 --warmup 20 --repeats 1 --no-cpu-baseline > $O/pmc_write.log 2>&1
python $R/tools/pmc_aggregate.py $O/pmc_hbm.json $O/pmc_fetch $O/pmc_write > /dev/null
rm -rf $O/pmc_fetch $O/pmc_write
cat $O/pmc_hbm.json | head -60
