#!/usr/bin/env python3
"""Retire time by phase (wave-us summed over all wavefronts of one launch), two launches vs the fused step (GPU box, profile build)."""
import json, os, sys
os.environ.setdefault("PCC_DEBUG_TIMELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcc_rl_amd
N = 65536
dev = torch.device("cuda:0")
names = {3: "state loads", 4: "boundary search", 5: "candidates+repairs", 6: "ending event", 7: "write-back",
         8: "rtt_means", 9: "metrics+batch loads", 10: "history+obs", 11: "outputs"}
res = {}
for mode, tune in (("two", dict(fused=0)), ("fused", dict(fused=1)), ("fused_serial", dict(fused=1, fused_debug=4))):
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
    env.set_tuning(**tune)
    gen = torch.Generator(device=dev).manual_seed(1234)
    acts = torch.rand((400, N), generator=gen, device=dev) * 2 - 1
    env.reset()
    rb = (N + 7) // 8 + 1
    for t in range(201):
        env.step(acts[t])
        if t == 199:
            prev = env.debug_timeline().astype(np.int64)[2 * N:2 * N + 2 * rb].reshape(-1, 16)
    cur = env.debug_timeline().astype(np.int64)[2 * N:2 * N + 2 * rb].reshape(-1, 16)
    bl = cur - prev
    r = {v: float(bl[:, k].sum()) / 100.0 for k, v in names.items()}
    r["total"] = sum(r.values())
    res[mode] = r
    env.close()
print(json.dumps(res, indent=1))
print("%-22s %10s %10s %10s" % ("phase (wave-us)", "two", "fused", "serial"))
for v in list(names.values()) + ["total"]:
    print("%-22s %10.0f %10.0f %10.0f" % (v, res["two"][v], res["fused"][v], res["fused_serial"][v]))
