#!/usr/bin/env python3
"""Send-launch timelines at the same steps of consecutive episodes of one handle (auto-reset in lockstep), GPU box only:
why is the second episode's send half slower than the first's?  PCC_DEBUG_TIMELINE=1 python tools/episode_timeline.py"""
import json, os, sys
os.environ.setdefault("PCC_DEBUG_TIMELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pcc_rl_amd

N = 65536
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N), generator=gen, device=dev) * 2 - 1
env.reset()
for t in range(1300):
    env.step_send(acts[t % 400])
    if t % 400 in (100, 300):
        raw = env.debug_timeline().astype(np.int64)
        n_items = int(env.debug_pass_stats(reset=False)["items"])
        tl = raw[:n_items].copy()
        envid = tl[:, 3] >> 16
        tl[:, 3] &= 0xFFFF
        closed, chain, serial = (tl[:, 7] >> 8) & 0xFFFF, (tl[:, 7] >> 24) & 0xFFFF, (tl[:, 7] >> 40) & 0xFFFF
        keep = tl[:, 0] > tl[:, 0].max() - 100000     # (this launch's items: 1 ms)
        tl, envid, closed, chain, serial = tl[keep], envid[keep], closed[keep], chain[keep], serial[keep]
        t0 = tl[:, 0].min()
        start, mid, fin = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0, (tl[:, 2] - t0) / 100.0
        hv = tl[:, 3] > 0
        dur = fin - start
        pct = lambda x: [round(float(np.percentile(x, p)), 1) for p in (50, 90, 99, 100)]
        order = np.argsort(-fin)[:4]
        print(json.dumps({"episode": t // 400, "step": t % 400, "items": int(len(tl)), "span_us": round(float(fin.max()), 1),
                          "finish": pct(fin), "start": pct(start), "packets": int(tl[:, 4].sum()),
                          "heavy": {"n": int(hv.sum()), "busy_us": round(float(dur[hv].sum())), "packets": int(tl[hv, 4].sum()),
                                    "ns_per_pk": round(float(1e3 * dur[hv].sum() / max(1, tl[hv, 4].sum())), 1)},
                          "light": {"n": int((~hv).sum()), "busy_us": round(float(dur[~hv].sum())), "packets": int(tl[~hv, 4].sum()),
                                    "rounds_end": pct(mid[~hv]) if (~hv).any() else None},
                          "slowest": [{"start": round(float(start[i]), 1), "rounds_end": round(float(mid[i]), 1), "fin": round(float(fin[i]), 1),
                                       "pk": int(tl[i, 4]), "largest": int(tl[i, 5] & 0xFFFFFFFF), "wp_envs": int(tl[i, 3]), "env": int(envid[i]),
                                       "closed": int(closed[i]), "chain": int(chain[i]), "serial": int(serial[i])} for i in order]}), flush=True)
        for i in order[:2]:
            e = int(envid[i])
            if int(tl[i, 3]) > 0:
                print("   env", e, {k: float(env.state(k).reshape(-1)[e].item()) for k in ("bw", "dl", "maxq", "queue_delay", "queue_time", "now", "run_dur", "rate", "lr")}, flush=True)
    env.step_retire()
