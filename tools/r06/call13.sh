#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c13
mkdir -p $O
cd $R
PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl.json 2> $O/tl.err
tail -3 $O/tl.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c13/tl.json"))
for s in d:
    if s["step"] in (100, 300):
        print("step", s["step"], "span", s["span_us"], "light running", s["light_items_running_at_us"], "wave running", s["wave_items_running_at_us"])
        for f in (s["longest_light_items_progress"] or [])[:3]:
            print("   ", f)
PY
