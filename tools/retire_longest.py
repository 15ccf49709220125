#!/usr/bin/env python3
"""Which workgroups of a retire launch take longest, and where their time goes (PCC_DEBUG_TIMELINE=1; GPU box only):
the phase stamps of tools/retire_phases.py for the 12 longest workgroups of one mid-episode launch, next to the median
of the ordinary (8 lanes per env) workgroups."""
import os, sys
os.environ.setdefault("PCC_DEBUG_TIMELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, pcc_rl_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
STEP = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N), generator=gen, device=dev) * 2 - 1
env.reset()
names = {3: "state", 4: "search", 5: "cand+rep", 6: "event", 7: "wb", 8: "rtt", 9: "metrics", 10: "hist", 11: "out"}
prev = None
for t in range(STEP + 1):
    env.step(acts[t])
    if t == STEP - 1:
        prev = env.debug_timeline().astype(np.int64)[2 * N:].reshape(-1, 16)
    if t == STEP:
        cur = env.debug_timeline().astype(np.int64)[2 * N:].reshape(-1, 16)
        d = cur - prev                      # the phase sums accumulate over the launches
        dur = (cur[:, 1] - cur[:, 0]) / 100.0
        ok = cur[:, 0] > cur[:, 0].max() - 100000
        idx = np.nonzero(ok)[0]
        st = (cur[:, 0] - cur[idx, 0].min()) / 100.0
        print("launch span %.1f us, %d workgroups" % (float(((cur[idx, 1] - cur[idx, 0].min()) / 100.0).max()), len(idx)))
        for b in idx[np.argsort(-dur[idx])][:12]:
            print("wg %5d start %5.1f dur %5.1f" % (b, st[b], dur[b]), {v: round(float(d[b, k]) / 100.0 / 2, 1) for k, v in names.items()})   # per wavefront (2 per workgroup)
        nb = idx[dur[idx] < 40]
        print("median of the workgroups under 40 us", {v: round(float(np.median(d[nb, k])) / 100.0 / 2, 1) for k, v in names.items()})
