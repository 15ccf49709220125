#!/bin/bash
# Counters of the fused step against the two launches over 200 bench steps: who waits, instruction cache, address path.
#   bash tools/pmc_fused.sh OUT.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/${1:-gpurun_out/pmc_fused.json}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -o "SQC_ICACHE[A-Z_]*\|SQ_IFETCH[A-Z_]*\|SQC_INST[A-Z_]*" | sort -u | tr "\n" " " > /tmp/icache_names.txt
cat /tmp/icache_names.txt; echo
i=0
for mode in fused two; do
  flag=""; [ $mode = two ] && flag="--two-launch"
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
             "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_LDS SQ_IFETCH" \
             "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
             "TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
             "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
             "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_ATOMIC_sum"; do
    rm -rf /tmp/pf_$i
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pf_$i -o t -- python $R/bench.py $flag --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-pmc --no-policy > /tmp/pf_$i.log 2>&1
    echo "$mode" > /tmp/pf_$i/mode
    i=$((i+1))
  done
done
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
out = collections.defaultdict(lambda: collections.defaultdict(dict))
for d in sorted(glob.glob("/tmp/pf_*")):
    try: mode = open(d + "/mode").read().strip()
    except OSError: continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "step_fused_kernel" if "step_fused" in n else "send_kernel" if "send_kernel" in n else "retire_kernel" if "retire_kernel" in n else None
            if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            if mode == "fused" and k != "step_fused_kernel": continue
            for c, v in agg[k].items():
                out[mode][k][c] = sum(v) / len(v)
            out[mode][k]["launches"] = len(v)
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
