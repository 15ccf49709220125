#!/usr/bin/env python3
"""Which items of a send launch are its critical path when the episode phases are staggered (`bench.py --stagger`)?
PCC_DEBUG_TIMELINE=1, GPU box only: the slowest items of one launch with the passes they took and their envs' state."""
import json, os, sys
os.environ.setdefault("PCC_DEBUG_TIMELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pcc_rl_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N), generator=gen, device=dev) * 2 - 1
env.reset()
phase = torch.arange(N, device=dev) % 400
for s in range(400):
    if s:
        env.reset(phase == s)
    env.step(acts[s % 400])
for t in range(400, 432):
    if t not in (410, 430):
        env.step(acts[t % 400])
        continue
    names = ("bw", "dl", "maxq", "queue_delay", "queue_time", "now", "run_dur", "rate", "lr", "steps")
    before = {k: env.state(k).reshape(-1).cpu().numpy().copy() for k in names}
    env.step_send(acts[t % 400])
    raw = env.debug_timeline().astype(np.int64)
    n_items = int(env.debug_pass_stats(reset=False)["items"])
    tl = raw[:n_items].copy()
    envid = tl[:, 3] >> 16
    tl[:, 3] &= 0xFFFF
    closed, chain, serial = (tl[:, 7] >> 8) & 0xFFFF, (tl[:, 7] >> 24) & 0xFFFF, (tl[:, 7] >> 40) & 0xFFFF
    t0 = tl[:, 0].min()
    start, fin = (tl[:, 0] - t0) / 100.0, (tl[:, 2] - t0) / 100.0
    order = np.argsort(-fin)[:14]
    print(json.dumps({"step": t, "items": n_items, "span_us": round(float(fin.max()), 1),
                      "finish_percentiles_50_90_99": [round(float(np.percentile(fin, p)), 1) for p in (50, 90, 99)]}))
    for i in order:
        e = int(envid[i])
        st = {k: float(before[k][e]) for k in names}
        print(json.dumps({"item": int(i), "start": round(float(start[i]), 1), "fin": round(float(fin[i]), 1), "pk": int(tl[i, 4]),
                          "wp_envs": int(tl[i, 3]), "closed": int(closed[i]), "chain": int(chain[i]), "serial": int(serial[i]),
                          "env": e, "state": {k: (round(v, 5) if k != "steps" else int(v)) for k, v in st.items()}}), flush=True)
    env.step_retire()
