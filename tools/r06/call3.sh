#!/bin/bash
# round 6, GPU call 3: the wave path's records staged through LDS (coalesced stores): parity, A/B against round 5's library, timeline
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c3
mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q > $O/parity.txt 2>&1
tail -5 $O/parity.txt
L=pcc-rl_amd/lib
timeout 900 python tools/ab_libraries.py 3 $L/libpcc_sim_r05.so $L/libpcc_sim.so > $O/ab_send.txt 2>&1
tail -3 $O/ab_send.txt
PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY=$R/$L/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl_new.json 2> $O/tl_new.err
python - <<'PY'
import json, glob, os
for f in sorted(glob.glob(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c3/tl_*.json")):
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
