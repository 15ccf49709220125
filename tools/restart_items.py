#!/usr/bin/env python3
"""What the restart items of the send launch cost (PCC_DEBUG_TIMELINE=1; GPU box only): episode phases
staggered like `bench.py --stagger`, then the timeline of one send launch -- the restart items are the
first n of the hand-out order."""
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
acts = torch.rand((64, N), generator=gen, device=dev) * 2 - 1
env.reset()
phase = torch.arange(N, device=dev) % 400
for s in range(400):
    if s:
        env.reset(phase == s)
    env.step(acts[s % 64])
out = []
for t in range(400, 440):
    env.step(acts[t % 64])
    if t in (410, 430):
        raw = env.debug_timeline().astype(np.int64)
        n_items = int(env.debug_pass_stats(reset=False)["items"])
        it = raw[:n_items]
        t0 = it[:, 0].min()
        st, en = (it[:, 0] - t0) / 100.0, (it[:, 2] - t0) / 100.0
        n_r = N // 400 + 1
        order = np.argsort(-(en - st))[:8]
        late = np.argsort(-en)[:8]
        busy = float((en - st).sum())
        out.append(dict(step=t, items=n_items, span_us=float(en.max()),
                        first_items=[dict(item=int(k), start=float(st[k]), end=float(en[k]), packets=int(it[k, 4]))
                                     for k in range(0, min(n_r, 6))],
                        busy_wave_us=busy, started_after_100us=int((st > 100).sum()), started_after_150us=int((st > 150).sum()),
                        latest=[dict(item=int(k), start=float(st[k]), end=float(en[k]), packets=int(it[k, 4]), wave_path_packets=int(it[k, 6])) for k in late],
                        longest=[dict(item=int(k), start=float(st[k]), end=float(en[k]), packets=int(it[k, 4]),
                                      wave_path_packets=int(it[k, 6])) for k in order]))
print(json.dumps(out, indent=1))
