#!/bin/bash
# Register / scratch / occupancy / spill figures of every kernel of the library, one line each (cross-compiles without a
# GPU): compiles each translation unit of pcc-rl_amd/csrc with -Rpass-analysis=kernel-resource-usage.  (pcc-rl_amd/build.py
# keeps the same report next to the built library as <lib>.resources.json; tests/test_abi_cpu.py asserts it.)
# usage: tools/resources.sh [extra hipcc flags, e.g. -DPCC_RETIRE_OCC=5]
R=$(cd "$(dirname "$0")/.." && pwd)
for U in $R/pcc-rl_amd/csrc/*.hip; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -I $R/include -I $R/pcc-rl_amd/csrc -c $U -o /tmp/pcc_res.o \
    -Rpass-analysis=kernel-resource-usage "$@" 2>&1
done | python3 -c '
import re, sys, subprocess
for line in sys.stdin:
    if "error" in line:
        print(line.rstrip()); continue
    m = re.search(r"remark:\s+(.*?)\s+\[-Rpass", line)
    if not m: continue
    t = m.group(1)
    if t.startswith("Function Name:"):
        name = subprocess.run(["c++filt", t.split(":",1)[1].strip()], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name).split("(")[0]
        print("\n%-36s" % name, end=" ")
    elif any(k in t for k in ("VGPRs:", "ScratchSize", "Occupancy", "LDS Size", "Spill")):
        print(re.sub(r"\s+", " ", t.replace(" [bytes/lane]", "").replace(" [waves/SIMD]", "").replace(" [bytes/block]", "")), end=" | ")
print()'
