mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for cfg in "512 0.45" "1024 0.45" "2048 0.45" "1024 0.6"; do
  set -- $cfg
  cd /tmp; PCC_HEAVY_PACKETS=$1 PCC_HEAVY_RHO=$2 timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sw_$1_$2 -o st -- python $R/tools/step_stats.py 65536 410 $R/gpurun_out/sw_$1_$2.json > /dev/null 2>&1
  cd $R
done
