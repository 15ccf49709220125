#!/bin/bash
# PMC passes over the send kernel for one case: tools/pmc_send.sh OUTDIR case waves steps
O=$1; CASE=$2; WAVES=$3; STEPS=${4:-60}
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_WR" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_ACTIVE_INST_SCA SQ_IFETCH SQ_INSTS_LDS" \
           "TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_FLAT_WRITE_WAVEFRONTS" \
           "TCP_PENDING_STALL_CYCLES TCP_TCC_WRITE_REQ TCP_LFIFO_STALL_CYCLES TCP_RFIFO_STALL_CYCLES" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/tools/light_only.py $CASE $WAVES $STEPS > $O/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
out = collections.OrderedDict()
for f in sorted(glob.glob("$O/p*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        if "send_kernel" not in k and "retire_kernel" not in k: continue
        a = acc[(k, r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), (s, n) in acc.items():
        out.setdefault(k, {})[c] = s / n
print(json.dumps(out, indent=1))
json.dump(out, open("$O/pmc_summary.json", "w"), indent=1)
PY
