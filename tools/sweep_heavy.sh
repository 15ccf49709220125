mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cfg in "256 2" "256 1" "256 3" "128 2" "512 2" "64 2"; do
  set -- $cfg
  cd /tmp; PCC_ROUND=$1 PCC_TAKEOVER=$2 timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sw_$1_$2 -o st -- python $R/tools/step_stats.py 65536 410 $R/gpurun_out/sw_$1_$2.json > /dev/null 2>&1
  cd $R
done
