mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for hp in 4096 6144 8192 1000000000; do
  cd /tmp; PCC_HEAVY_PREDICT=$hp timeout 120 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/sw_hp$hp -o st -- python $R/tools/step_stats.py 65536 410 $R/gpurun_out/sw_hp$hp.json > /dev/null 2>&1
  cd $R
done
