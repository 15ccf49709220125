#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c10
mkdir -p $O
cd $R
for rp in 32 64 128 192; do
PCC_TUNE_ROUND_PACKETS=$rp PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl_rp$rp.json 2> $O/tl_rp$rp.err
done
python - <<'PY'
import json, os
for rp in (32, 64, 128, 192):
    d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c10/tl_rp%d.json" % rp))
    for s in d:
        if s["step"] in (100, 300):
            L = [w for w in s["slowest"] if w["first_round_us"]]
            print("round_packets", rp, "step", s["step"], "span", s["span_us"], [(w["start"], w["first_round_us"], w["finish"], w["largest_env"]) for w in L[:4]])
PY
