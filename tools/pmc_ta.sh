#!/bin/bash
# Address-path counters of the two kernels (GPU box, from the repo root): how busy the TA is, how often its
# FIFOs are full, what stalls it, the L1's read-request latency.  bash tools/pmc_ta.sh OUT.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/pmc_ta.json}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum" \
           "TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_READ_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"; do
  rm -rf /tmp/ta_$i
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/ta_$i -o t -- python $R/bench.py --steps 60 --warmup 20 --repeats 1 --no-cpu-baseline --no-policy > /tmp/ta_$i.log 2>&1
  i=$((i+1))
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = collections.defaultdict(dict)
for d in sorted(glob.glob("/tmp/ta_*")):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "send_kernel" if "send_kernel" in n else "retire_kernel" if "retire_kernel" in n else None
            if k:
                agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            for c, v in agg[k].items():
                out[k][c + "_mean_per_launch"] = sum(v) / len(v)
            out[k]["launches"] = len(v)
out["_note"] = "rocprofv3 --pmc (one small set per pass, --kernel-trace only) over `python bench.py --steps 60 --warmup 20 --repeats 1 --no-cpu-baseline` (steps 20..80 of an episode); means per launch"
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
