#!/bin/bash
# Counter-level stall attribution of the two product kernels (round 6, verdict item 2): SQ wait / issue / active split,
# instruction mix, TCP / TA / TCC / address-translation counters -- one small set per rocprofv3 pass (--pmc with
# --kernel-trace only), over `bench.py --steps 200` each; at the default ring pools (85 GB span) and at the floor
# divisors 2,8,32 (6.4 GB).  Counter names are intersected with what `rocprofv3 -L` lists on this box.
#   bash tools/r06/pmc_attrib.sh OUT_DIR
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/${1:-gpurun_out/r06_pmc}
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail_raw.txt 2>&1
python3 - "$O" <<'PY'
import re, sys
o = sys.argv[1]
txt = open(o + "/counters_avail_raw.txt", errors="replace").read()
names = sorted(set(re.findall(r"\b((?:SQ|SQC|TCP|TA|TD|TCC|TCA|GRBM|CPC|CPF|SPI|GL2C|UTCL2|ATC|MC|EA)[A-Z0-9_]*_[A-Za-z0-9_]+)\b", txt)))
open(o + "/counters_avail.txt", "w").write("\n".join(names) + "\n")
want = [
 ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_VMEM"],
 ["SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_MISC", "SQ_ACTIVE_INST_FLAT", "SQ_INST_CYCLES_VMEM_RD", "SQ_INST_CYCLES_VMEM_WR", "SQ_INST_CYCLES_SMEM", "SQ_WAIT_INST_LDS"],
 ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM", "SQ_INSTS_LDS", "SQ_INSTS_FLAT", "SQ_INSTS_BRANCH"],
 ["SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_SMEM", "SQ_INST_LEVEL_LDS", "SQ_LEVEL_WAVES", "SQ_WAVES_EQ_64", "SQ_WAVES_LT_64", "SQ_ACTIVE_INST_EXP_GDS", "SQ_THREAD_CYCLES_VALU"],
 ["TCP_PENDING_STALL_CYCLES_sum", "TCP_TCC_READ_REQ_LATENCY_sum", "TCP_TCC_READ_REQ_sum", "TCP_TCC_WRITE_REQ_sum"],
 ["TCP_TCC_WRITE_REQ_LATENCY_sum", "TCP_TCC_ATOMIC_WITH_RET_REQ_sum", "TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum", "TCP_TOTAL_CACHE_ACCESSES_sum"],
 ["TCP_UTCL1_REQUEST_sum", "TCP_UTCL1_TRANSLATION_MISS_sum", "TCP_UTCL1_TRANSLATION_HIT_sum", "TCP_UTCL1_PERMISSION_MISS_sum"],
 ["TCP_TA_TCP_STATE_READ_sum", "TCP_TCP_TA_DATA_STALL_CYCLES_sum", "TCP_TD_TCP_STALL_CYCLES_sum", "TCP_TCR_TCP_STALL_CYCLES_sum"],
 ["TCP_GATE_EN1_sum", "TCP_GATE_EN2_sum", "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum", "TCP_WRITE_TAGCONFLICT_STALL_CYCLES_sum"],
 ["TCP_TOTAL_ACCESSES_sum", "TCP_TOTAL_READ_sum", "TCP_TOTAL_WRITE_sum", "TCP_TOTAL_ATOMIC_WITH_RET_sum"],
 ["TA_BUSY_avr", "TA_BUSY_max", "TA_TA_BUSY_sum", "TA_ADDR_STALLED_BY_TC_CYCLES_sum"],
 ["TA_DATA_STALLED_BY_TC_CYCLES_sum", "TA_ADDR_STALLED_BY_TD_CYCLES_sum", "TA_BUFFER_WAVEFRONTS_sum", "TA_FLAT_WAVEFRONTS_sum"],
 ["TCC_HIT_sum", "TCC_MISS_sum", "TCC_REQ_sum", "TCC_ATOMIC_sum"],
 ["TCC_EA0_RDREQ_sum", "TCC_EA0_RD_UNCACHED_32B_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_WRREQ_sum"],
 ["TCC_EA0_RDREQ_LEVEL_sum", "TCC_EA0_WRREQ_LEVEL_sum", "TCC_EA0_WRREQ_64B_sum", "TCC_EA0_WRREQ_STALL_sum"],
 ["TCC_TAG_STALL_sum", "TCC_BUSY_sum", "TCC_READ_sum", "TCC_WRITE_sum"],
 ["GRBM_GUI_ACTIVE", "GRBM_COUNT"],
]
have = set(names)
sets = []
for s in want:
    ok = [c for c in s if c in have]
    if ok:
        sets.append(" ".join(ok))
missing = [c for s in want for c in s if c not in have]
open(o + "/sets.txt", "w").write("\n".join(sets) + "\n")
open(o + "/missing.txt", "w").write("\n".join(missing) + "\n")
print(len(names), "counters listed;", len(sets), "sets;", "missing:", " ".join(missing))
PY
i=0
for pools in default floor; do
  export PCC_BENCH_RING_POOLS=""
  [ $pools = floor ] && export PCC_BENCH_RING_POOLS="2,8,32"
  while read -r set; do
    [ -z "$set" ] && continue
    d=/tmp/pa_${pools}_$i
    rm -rf $d
    timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $d -o t -- python $R/bench.py --steps 200 --warmup 20 --repeats 1 --no-cpu-baseline --no-pmc --no-policy > $d.log 2>&1
    rc=$?
    echo "$pools" > $d/pools 2>/dev/null
    echo "$pools set $i rc=$rc  $set"
    i=$((i+1))
    # the floor pools only for the sets that decide the question (translation, TCP stalls, SQ waits): see below
  done < <(if [ $pools = default ]; then cat $O/sets.txt; else grep -E "SQ_WAIT_ANY|TCP_PENDING|UTCL1|TCC_HIT|TCP_TD_TCP|TCC_EA0_RDREQ_LEVEL" $O/sets.txt; fi)
done
python3 - "$O" <<'PY'
import collections, csv, glob, json, os, sys
o = sys.argv[1]
out = collections.defaultdict(lambda: collections.defaultdict(dict))
for d in sorted(glob.glob("/tmp/pa_*")):
    if not os.path.isdir(d):
        continue
    try: pools = open(d + "/pools").read().strip()
    except OSError: continue
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "send_kernel" if "send_kernel" in n else "retire_kernel" if "retire_kernel" in n else None
            if k: agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k in agg:
            for c, v in agg[k].items():
                v = v[20:] if len(v) > 60 else v     # (the warm-up steps)
                out[pools][k][c] = sum(v) / len(v)
                out[pools][k]["launches"] = len(v)
    # kernel durations of the profiled pass (ns)
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            k = "send_kernel" if "send_kernel" in n else "retire_kernel" if "retire_kernel" in n else None
            if k: dur[k].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
        for k, v in dur.items():
            out[pools][k].setdefault("duration_us_by_pass", []).append(round(sum(v) / len(v) / 1e3, 2))
json.dump(out, open(o + "/pmc_attrib.json", "w"), indent=1, sort_keys=True)
print(json.dumps(out, indent=1, sort_keys=True)[:6000])
PY
