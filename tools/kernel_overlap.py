#!/usr/bin/env python3
"""Do the kernels of the send half overlap?  Reads a `rocprofv3 --kernel-trace --output-format csv` directory and prints, per
kernel name, the launch count and average duration, and for the send half of each step how the light / wave / restart
kernels sit in time (start offsets and ends relative to the first of them, and the retire kernel behind them).
usage: kernel_overlap.py DIR [first_step last_step]"""
import csv, glob, json, os, sys
d = sys.argv[1]
paths = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for p in paths:
    with open(p) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
short = lambda n: n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "").split("(")[0]
agg = {}
for s, e, n in rows:
    a = agg.setdefault(short(n), [0, 0.0])
    a[0] += 1; a[1] += (e - s) / 1e3
print(json.dumps({k: {"launches": v[0], "avg_us": round(v[1] / v[0], 2)} for k, v in agg.items()}, indent=1))
# steps: a retire_kernel<.., false> launch closes a step; the send kernels before it belong to it
steps, cur = [], []
for s, e, n in rows:
    k = short(n)
    if k.startswith("send_") or k.startswith("retire_kernel"):
        cur.append((s, e, k))
        if k.startswith("retire_kernel"):
            steps.append(cur); cur = []
lo = int(sys.argv[2]) if len(sys.argv) > 2 else len(steps) // 2
hi = int(sys.argv[3]) if len(sys.argv) > 3 else lo + 8
for st in steps[lo:hi]:
    t0 = min(s for s, e, k in st)
    print(" | ".join("%s %.1f..%.1f" % (k.replace("_kernel", "").replace("<1, false>", ""), (s - t0) / 1e3, (e - t0) / 1e3) for s, e, k in st))
span = [(max(e for s, e, k in st if k.startswith("send_")) - min(s for s, e, k in st if k.startswith("send_"))) / 1e3 for st in steps if any(k.startswith("send_") for s, e, k in st)]
if span:
    print("send half span (first start .. last end) avg us:", round(sum(span) / len(span), 2), "over", len(span), "steps")
