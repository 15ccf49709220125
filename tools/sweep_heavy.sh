mkdir -p gpurun_out; export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT
for sk in 0 1 2; do
  cd /tmp; PCC_DEBUG_SKIP=$sk timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace --output-format csv -d $R/gpurun_out/pq_$sk -o q -- python $R/tools/step_stats.py 65536 60 $R/gpurun_out/tmp.json > /dev/null 2>&1
  cd $R
done
