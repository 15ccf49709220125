#!/bin/bash
# Calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE against kernels of known traffic (tools/microbench/pmc_calib.hip).
# Run on the GPU box from the repo root:  bash tools/pmc_calibrate.sh OUT.json
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=${1:-$R/gpurun_out/pmc_calibration.json}
B=$R/tools/microbench/pmc_calib
[ -x $B ] || hipcc --offload-arch=gfx950 -O3 $R/tools/microbench/pmc_calib.hip -o $B || exit 1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/calib_f /tmp/calib_w
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/calib_f -o f -- $B > /tmp/calib_known.json 2> /tmp/calib_f.err
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/calib_w -o w -- $B > /dev/null 2> /tmp/calib_w.err
python - "$OUT" <<'PY'
import collections, csv, glob, json, sys
known = json.loads([l for l in open("/tmp/calib_known.json") if l.startswith("{")][-1])
got = collections.defaultdict(lambda: collections.defaultdict(list))
for d in ("/tmp/calib_f", "/tmp/calib_w"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            for k in known:
                if k in r["Kernel_Name"]:
                    got[k][r["Counter_Name"]].append(float(r["Counter_Value"]) * 1024.0)
out = {}
for k, kb in known.items():
    e = dict(requested=kb)
    for c, key in (("FETCH_SIZE", "read"), ("WRITE_SIZE", "write")):
        v = got[k].get(c, [])
        if not v:
            continue
        e[c + "_bytes"] = sum(v) / len(v)
        if kb[key]:
            e[c + "_over_requested"] = e[c + "_bytes"] / kb[key]
        for alt in ("lines_x128", "records_x16"):
            if alt in kb and kb[key]:
                e[c + "_over_" + alt] = e[c + "_bytes"] / kb[alt]
    out[k] = e
out["_note"] = ("counter bytes (KB column x 1024, mean of 2 launches) over the bytes each kernel requests; lines_x128 = every touched "
                "128-byte line counted whole, records_x16 = every touched 16-byte record counted whole")
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
PY
