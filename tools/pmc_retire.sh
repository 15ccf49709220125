cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_ANY SQ_WAVES"; do
  tag=$(echo $set | tr ' ' '_' | cut -c1-40)
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$tag -- python $R/bench.py --steps 60 --warmup 20 --no-cpu-baseline --split > $R/gpurun_out/pmc_$tag.log 2>&1
  echo "$tag rc=$?"
done
ls $R/gpurun_out/pmc_*
