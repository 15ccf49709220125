#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c20
mkdir -p $O
cd $R
PCC_TL_ACTIONS=randn PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl_randn.json 2> $O/tl.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c20/tl_randn.json"))
for s in d:
    if s["step"] in (10, 30, 60, 100, 200, 300):
        cp = s["critical_path"]; li = s["light_items"]; hv = s["heavy_items"]
        print("step", s["step"], "span", s["span_us"], "pk", s["packets_total"], "wave pk", s["wave_path_packets"], "finish", [round(x,1) for x in s["finish_us"]], "longest light", round(cp["longest_light_item"]["us"],1), "longest heavy", cp["longest_heavy_item"], "heaviest", cp["heaviest_env"], "heavy ns/pkt", round(hv["ns_per_packet"],1), "n heavy", hv["n"])
        for w in s["slowest"][:3]: print("     ", {k: w[k] for k in ("start","finish","packets","wave_path_envs","largest_env")})
PY
