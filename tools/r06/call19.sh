#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/r06_c19
mkdir -p $O
cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "two_sender or philox_batches or send_paths or golden_vectors or out_of_lockstep or team or wave_path" > $O/parity.txt 2>&1; tail -2 $O/parity.txt
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-policy --no-scaling --config 5 --steps 800 --repeats 1 > $O/bench_config5_$i.log 2>&1
python -c "
import json,sys
d=json.loads([l for l in open('$O/bench_config5_$i.log') if l.startswith('{')][-1]); r=d['roofline']
print('config5 value %.4g ms %.4f send %.4f retire %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['other_kernels'][0]['kernel_ms']))"
timeout 300 python bench.py --no-cpu-baseline --no-pmc --no-policy --no-scaling --steps 1200 --repeats 1 > $O/bench_config3_$i.log 2>&1
python -c "
import json,sys
d=json.loads([l for l in open('$O/bench_config3_$i.log') if l.startswith('{')][-1]); r=d['roofline']
print('config3 value %.4g ms %.4f send %.4f retire %.4f' % (d['value'], d['ms_per_step'], r['kernel_ms'], r['other_kernels'][0]['kernel_ms']))"
done
PCC_DEBUG_TIMELINE=1 PCC_SIM_LIBRARY=$R/pcc-rl_amd/lib/libpcc_sim_prof.so timeout 300 python tools/send_timeline.py 32768 2 > $O/tl_c5.json 2> $O/tl_c5.err
python - <<'PY'
import json, os
d = json.load(open(os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/r06_c19/tl_c5.json"))
for s in d:
    if s["step"] in (100, 200, 300):
        cp = s["critical_path"]
        print("step", s["step"], "span", s["span_us"], "finish p50/p90/p99", [round(x,1) for x in s["finish_us"][:3]], "longest light", round(cp["longest_light_item"]["us"],1), "longest heavy", cp["longest_heavy_item"])
        for w in s["slowest"][:4]: print("     ", {k: w[k] for k in ("start","finish","packets","wave_path_envs")})
PY
