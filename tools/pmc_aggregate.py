#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files -> one JSON (profiles/*_pmc_hbm.json).

    python tools/pmc_aggregate.py out.json DIR_WITH_FETCH_SIZE DIR_WITH_WRITE_SIZE
"""
import collections, csv, glob, json, sys

def short(name):
    for k in ("step_kernel", "send_kernel", "retire_kernel", "reset_init_kernel"):
        if k in name:
            targs = name[name.index(k) + len(k):].split(">")[0].lstrip("<")
            return "%s<%s>" % (k, targs)
    return None

out = {}
for d in sys.argv[2:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        seen = collections.defaultdict(set)
        for r in csv.DictReader(open(f)):
            k = short(r["Kernel_Name"])
            if not k:
                continue
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            seen[k].add(r["Dispatch_Id"])
        for k in agg:
            e = out.setdefault(k, {})
            e["launches"] = len(seen[k])
            for c, v in agg[k].items():
                e[c + "_KB_mean_per_launch"] = v / len(seen[k])
for k, e in out.items():
    if "FETCH_SIZE_KB_mean_per_launch" in e and "WRITE_SIZE_KB_mean_per_launch" in e:
        e["hbm_bytes_per_launch_raw"] = 1024.0 * (e["FETCH_SIZE_KB_mean_per_launch"] + e["WRITE_SIZE_KB_mean_per_launch"])
out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 100 --warmup 20 "
                "--no-cpu-baseline` (steps 20..120 of an episode); counters are KB per dispatch.  MI355X_MICROARCH.md: on gfx950 "
                "FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read and is uncalibrated for other widths; these kernels "
                "touch 16-B records at scattered addresses, so the raw values are reported uncorrected.")
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps(out, indent=1))
