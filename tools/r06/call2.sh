#!/bin/bash
# round 6, GPU call 2: what bounds a lane-round iteration?  (a) the microbenchmark with FEW probe wavefronts (one per 8 compute
# units, one per compute unit): the coding's own speed, no chip-wide store rate in the way; (b) the send launch's per-item
# timeline (profile builds) of round 5's sources against the branch-free pipelined lane rounds, and with the lane rounds'
# stores / Philox skipped (timing only).
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c2
mkdir -p $O
cd $R
for np in 32 256; do timeout 120 tools/microbench/lane_round 0 $np; done > $O/lane_round_few.txt 2>&1
cat $O/lane_round_few.txt
L=$R/pcc-rl_amd/lib
export PCC_DEBUG_TIMELINE=1
PCC_SIM_LIBRARY=$L/libpcc_sim_prof_r05.so timeout 300 python tools/send_timeline.py > $O/tl_r05.json 2> $O/tl_r05.err
PCC_SIM_LIBRARY=$L/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl_new.json 2> $O/tl_new.err
PCC_DEBUG_SKIP=4 PCC_SIM_LIBRARY=$L/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl_new_nostore.json 2> $O/tl_new_nostore.err
PCC_DEBUG_SKIP=12 PCC_SIM_LIBRARY=$L/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl_new_nostore_nophilox.json 2> $O/tl_new_nsnp.err
PCC_DEBUG_SKIP=4 PCC_SIM_LIBRARY=$L/libpcc_sim_prof_r05.so timeout 300 python tools/send_timeline.py > $O/tl_r05_nostore.json 2> $O/tl_r05_nostore.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c2/tl_*.json")):
    try: d = json.load(open(f))
    except Exception as e: print(f, "unreadable", e); continue
    print(os.path.basename(f))
    for s in d:
        if s["step"] in (2, 100, 200, 300):
            cp = s["critical_path"]; li = s["light_items"]; hv = s["heavy_items"]
            print("  step %3d span %.1f  longest light %.1f us (%d it, %.0f ns/it)  light mean %.0f ns/it  heavy %.1f ns/pkt  longest heavy %.1f  finish p50/p90/p99 %s" % (
                s["step"], s["span_us"], cp["longest_light_item"]["us"], cp["longest_light_item"]["lane_iterations"], cp["longest_light_item"]["ns_per_iteration"],
                li["ns_per_iteration"], hv["ns_per_packet"], cp["longest_heavy_item"]["us"], [round(x, 1) for x in s["finish_us"][:3]]))
PY
