#!/usr/bin/env python3
"""Per-kernel means of rocprofv3 --pmc counter_collection.csv files -> one JSON (profiles/*_pmc_hbm.json).

    python tools/pmc_aggregate.py out.json DIR_WITH_FETCH_SIZE DIR_WITH_WRITE_SIZE

bench.py imports aggregate() for the counter passes it runs over itself.
"""
import collections, csv, glob, json, os, sys


def short(name):
    for k in ("step_small_kernel", "step_kernel", "send_restart_kernel", "send_light_kernel", "send_wave_kernel", "refill_kernel", "send_kernel", "retire_kernel", "reset_init_kernel"):
        if k in name:
            targs = name[name.index(k) + len(k):].split(">")[0].lstrip("<")
            return "%s<%s>" % (k, targs)
    return None


def aggregate(dirs, bench_line=None):
    """dirs: rocprofv3 output directories (one per counter pass); bench_line: the JSON line the profiled bench.py printed
    (its roofline names the window's algorithmic bytes), or None to look for DIR.log next to the first directory."""
    out = {}
    for d in dirs:
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
            # calibration (profiles/r02_pmc_calibration.json): FETCH_SIZE counts every fetched 128-byte line as 64 bytes, for
            # dense 16-byte reads, one 16-byte record per line and the 8-byte strided ring reads alike -> x 2;
            # WRITE_SIZE is exact for dense writes and counts 32-byte sectors for lone 16-byte records (ring appends: 1.12 x)
            e["hbm_bytes_per_launch"] = 1024.0 * (2.0 * e["FETCH_SIZE_KB_mean_per_launch"] + e["WRITE_SIZE_KB_mean_per_launch"])
    # the bench line of the profiled run names the window's algorithmic bytes
    if bench_line is None:
        for d in dirs:
            log = d.rstrip("/") + ".log"
            if os.path.exists(log):
                lines = [l for l in open(log) if l.startswith("{")]
                if lines:
                    bench_line = lines[-1]
                    break
    if bench_line:
        r = json.loads(bench_line).get("roofline", {})
        for k in [r] + r.get("other_kernels", []):
            # (match on the kernel's base name: the template arguments in the bench line may lag the code)
            name = next((n for n in out if n.split("<")[0] == str(k.get("kernel", "")).split("<")[0] and n.split("<")[1][:1] == str(k.get("kernel", "<1")).split("<")[1][:1]), None)
            if name and "algorithmic_bytes_per_launch" in k:
                out[name]["algorithmic_bytes_per_launch"] = k["algorithmic_bytes_per_launch"]
                if "hbm_bytes_per_launch" in out[name]:
                    out[name]["traffic_over_algorithmic"] = out[name]["hbm_bytes_per_launch"] / k["algorithmic_bytes_per_launch"]
    out["_note"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `python bench.py --steps 400 --warmup 20 "
                    "--repeats 1 --no-cpu-baseline` (one whole episode); counters are KB per dispatch.  hbm_bytes_per_launch = "
                    "2 x FETCH_SIZE + WRITE_SIZE: tools/pmc_calibrate.sh (profiles/r02_pmc_calibration.json) measured FETCH_SIZE at "
                    "exactly half of the fetched 128-byte lines for every read pattern of these kernels and WRITE_SIZE at 1.0 x (dense) "
                    "to 1.12 x (ring appends) of the written bytes; hbm_bytes_per_launch_raw is the uncorrected sum.")
    out["_window"] = os.environ.get("PCC_PMC_WINDOW", "one whole 400-step episode (after 20 warm-up steps)")
    out["_commit"] = os.environ.get("PCC_COMMIT", "unknown")
    return out


if __name__ == "__main__":
    res = aggregate(sys.argv[2:])
    json.dump(res, open(sys.argv[1], "w"), indent=1)
    print(json.dumps(res, indent=1))
