#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c27
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "not noise and not cwnd" > $O/parity.txt 2>&1; tail -4 $O/parity.txt
L=pcc-rl_amd/lib
timeout 900 python tools/ab_libraries.py 4 $L/libpcc_sim_pos4.so $L/libpcc_sim.so > $O/ab.txt 2>&1; tail -1 $O/ab.txt
PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY=$R/$L/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl.json 2> $O/tl.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c27/tl.json"))
for s in d:
    if s["step"] in (100, 200, 300):
        cp = s["critical_path"]; hv = s["heavy_items"]
        print("step", s["step"], "span", s["span_us"], "finish", [round(x,1) for x in s["finish_us"]], "longest light", round(cp["longest_light_item"]["us"],1), "heavy ns/pkt", round(hv["ns_per_packet"],1), "longest heavy", round(cp["longest_heavy_item"]["us"],1), "wave running", s["wave_items_running_at_us"])
PY
