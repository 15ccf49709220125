#!/usr/bin/env python3
"""Light items of one step, fused against two launches (GPU box only, profile build): start, end of the lane rounds, end,
iterations of the longest lane, ns per iteration.  PCC_DEBUG_TIMELINE=1 python tools/fused_light_items.py [step]"""
import json, os, sys
os.environ.setdefault("PCC_DEBUG_TIMELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcc_rl_amd
N = 65536
STEP = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda:0")
res = {}
for mode in ("two", "fused"):
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
    env.set_tuning(fused=1 if mode == "fused" else 0)
    for k, v in os.environ.items():
        if k.startswith("PCC_TUNE_"):
            env.set_tuning(**{k[9:].lower(): float(v)})
    gen = torch.Generator(device=dev).manual_seed(1234)
    acts = torch.rand((400, N, 1), generator=gen, device=dev) * 2 - 1
    env.reset()
    for t in range(STEP):
        env.step(acts[t])
    if mode == "two":
        env.step_send(acts[STEP])
    else:
        env.step(acts[STEP])
    raw = env.debug_timeline().astype(np.int64)
    items = raw[:2 * N]
    it = items[items[:, 0] > 0]
    if mode == "fused":
        it = it[it[:, 0] > it[:, 0].max() - 100000]   # this launch's (1 ms window)
    t0 = it[:, 0].min()
    light = it[it[:, 3] == 0]
    st, mid, fin = (light[:, 0] - t0) / 100.0, (light[:, 1] - t0) / 100.0, (light[:, 2] - t0) / 100.0
    iters = np.maximum(1, light[:, 5])
    nspi = 1e3 * (mid - st) / iters
    order = np.argsort(-iters)
    pct = lambda x: [round(float(np.percentile(x, p)), 2) for p in (10, 50, 90, 99, 100)]
    res[mode] = {"light_items": int(len(light)), "start_us": pct(st), "rounds_us": pct(mid - st), "tail_us": pct(fin - mid), "end_us": pct(fin),
                 "iterations": pct(iters), "ns_per_iteration": pct(nspi),
                 "longest": [{"iters": int(iters[i]), "start": float(st[i]), "rounds_end": float(mid[i]), "end": float(fin[i]), "ns_per_iter": float(nspi[i])} for i in order[:6]],
                 "shortest": [{"iters": int(iters[i]), "start": float(st[i]), "rounds_end": float(mid[i]), "end": float(fin[i]), "ns_per_iter": float(nspi[i])} for i in order[-4:]]}
    if mode == "two":
        env.step_retire()
    env.close()
print(json.dumps(res, indent=1))
