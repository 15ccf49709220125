#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c11
mkdir -p $O
cd $R
(timeout 120 tools/microbench/lane_round 0 1024 1; timeout 120 tools/microbench/lane_round 1 1024 1) > $O/lane_round_mixed.txt 2>&1
cat $O/lane_round_mixed.txt
PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl.json 2> $O/tl.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c11/tl.json"))
for s in d:
    if s["step"] in (100, 300):
        print("step", s["step"], [(w["start"], w["first_round_us"], w["finish"], w["largest_env"], round(w["shader_clock_ghz"] or 0, 2)) for w in s["slowest"] if w["first_round_us"]][:5])
PY
