#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c18
mkdir -p $O
cd $R
PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py 32768 2 > $O/tl_c5.json 2> $O/tl_c5.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c18/tl_c5.json"))
for s in d:
    if s["step"] in (100, 200, 300):
        cp = s["critical_path"]; li = s["light_items"]; hv = s["heavy_items"]
        print("step", s["step"], "span", s["span_us"], "finish p50/p90/p99", [round(x,1) for x in s["finish_us"][:3]], "longest light", cp["longest_light_item"], "longest heavy", cp["longest_heavy_item"], "heaviest env", cp["heaviest_env"])
        print("   light", {k: (round(v,1) if isinstance(v,float) else v) for k,v in li.items()}, "heavy", {k: (round(v,1) if isinstance(v,float) else v) for k,v in hv.items() if k!="ns_per_packet_p10_p50_p90"}, hv["ns_per_packet_p10_p50_p90"])
        print("   light running", s["light_items_running_at_us"], "wave running", s["wave_items_running_at_us"])
        for w in s["slowest"][:6]: print("     ", w)
PY
