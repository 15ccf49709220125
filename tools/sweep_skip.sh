mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for sk in 0 1 2 3; do
  cd /tmp; PCC_DEBUG_SKIP=$sk timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sk_$sk -o st -- python $R/tools/step_stats.py 65536 120 $R/gpurun_out/sk_$sk.json > /dev/null 2>&1
  cd $R
done
