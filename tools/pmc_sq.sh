#!/bin/bash
# SQ counters of the two kernels of a step over one bench episode: who waits, who issues.  tools/pmc_sq.sh OUTDIR
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/$1
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INST_CYCLES_SALU" \
           "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -o p -- python $R/bench.py --steps 400 --warmup 20 --repeats 1 --no-cpu-baseline --no-pmc --no-policy > $O/p$i.log 2>&1
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
        out[k]["launches"] = n
print(json.dumps(out, indent=1))
json.dump(out, open("$O/pmc_sq.json", "w"), indent=1)
PY
