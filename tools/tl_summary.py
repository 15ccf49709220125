#!/usr/bin/env python3
"""One line per sampled step of a tools/send_timeline.py output: launch span, busy wavefront time, when the wavefronts
finish, the wave-path items (count, ns per packet, the longest) and the light items (count, ns per lane iteration, the longest).
usage: tl_summary.py timeline.json [...]"""
import json, sys
for path in sys.argv[1:]:
    try:
        d = json.load(open(path))
    except Exception as e:
        print(path, "no data:", e)
        continue
    for r in d:
        c, h, l = r["critical_path"], r["heavy_items"], r["light_items"]
        lh, ll = c["longest_heavy_item"], c["longest_light_item"]
        print(path.split("/")[-1], "step", r["step"], "span %.0f us" % r["span_us"], "busy %.0f ms" % (r["busy_wave_us_total"] / 1e3),
              "finish p50/p90/p99", [round(x) for x in r["finish_us"][:3]],
              "| wave items", h["n"], "ns/pk %.0f" % h["ns_per_packet"], "longest", lh and (round(lh["us"]), lh["packets"]),
              "| light items", l["n"], "ns/it %.0f" % l["ns_per_iteration"], "longest", ll and (round(ll["us"]), ll["lane_iterations"]))
