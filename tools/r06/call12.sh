#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c12
mkdir -p $O
cd $R
timeout 600 python tools/ab_knob.py '[{"light_wgs":40},{"light_wgs":0},{"light_wgs":40},{"light_wgs":32},{"light_wgs":24},{"light_wgs":28}]' 65536 3 50 > $O/sweep_lw.txt 2>&1
cat $O/sweep_lw.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "philox_batches or golden_vectors or send_paths or launch_shape" > $O/parity.txt 2>&1; tail -2 $O/parity.txt
PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py > $O/tl.json 2> $O/tl.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c12/tl.json"))
for s in d:
    if s["step"] in (2, 100, 200, 300):
        cp = s["critical_path"]
        print("step", s["step"], "span", s["span_us"], "start p50/p90/p99/max", [round(x,1) for x in s["start_us_p50_p90_p99_max"]], "longest light", cp["longest_light_item"]["us"], "longest heavy", cp["longest_heavy_item"]["us"], "light running", s["light_items_running_at_us"])
        print("    ", [(w["start"], w["first_round_us"], w["finish"], w["largest_env"]) for w in s["slowest"]][:5])
PY
