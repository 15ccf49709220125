#!/bin/bash
# Register / scratch / occupancy figures of every kernel in pcc_sim.hip (cross-compiles without a GPU).
R=$(cd "$(dirname "$0")/.." && pwd)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I $R/include $R/pcc-rl_amd/csrc/pcc_sim.hip \
  -o ${1:-/tmp/pcc_res.so} -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c '
import re, sys, subprocess
cur = None
for line in sys.stdin:
    if "error" in line or "warning:" in line:
        print(line.rstrip()); continue
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":",1)[1].strip()], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        print("\n%-28s" % name, end=" ")
    elif any(k in t for k in ("VGPRs:", "AGPRs", "ScratchSize", "Occupancy", "LDS Size", "SGPRs:")) and "Spill" not in t:
        print(re.sub(r"\s+", " ", t.replace(" [bytes/lane]", "").replace(" [waves/SIMD]", "").replace(" [bytes/block]", "")), end=" | ")
print()'
