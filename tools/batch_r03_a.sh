#!/bin/bash
# round-3 first GPU batch: baseline of the round-2 code, pipelined-halves prototype, knob sweep
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r03_a
mkdir -p $O
cd $R
timeout 200 python bench.py --steps 400 --warmup 50 --repeats 2 --no-cpu-baseline > $O/bench_base.log 2> $O/bench_base.err
timeout 300 python tools/pipeline_proto.py 65536 400 one par pipe > $O/pipeline_proto.log 2> $O/pipeline_proto.err
timeout 400 python tools/sweep2.py '[{"heavy_predict":512},{"heavy_predict":384},{"heavy_predict":256},{"heavy_predict":192},{"heavy_predict":512,"send_waves":12},{"heavy_predict":512,"send_waves":8}]' > $O/sweep.log 2> $O/sweep.err
tail -n 3 $O/bench_base.log | cut -c 1-1500
cat $O/pipeline_proto.log
cat $O/sweep.log
