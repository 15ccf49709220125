#!/usr/bin/env python3
"""Shape of one retire launch (PCC_DEBUG_TIMELINE=1; GPU box only): span of the launch, time of its
workgroups, how many run at a time, and which workgroups are still running at the end."""
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
out = []
for t in range(310):
    env.step(acts[t % 64])
    if t in (20, 100, 200, 300):
        raw = env.debug_timeline().astype(np.int64)
        bl = raw[2 * N:].reshape(-1, 16)            # one row per retire workgroup (behind the 2 n item slots of the send half)
        bl = bl[bl[:, 0] > bl[:, 0].max() - 100000]   # (workgroups without envs in this launch keep the stamps of an earlier one)
        st, en = bl[:, 0] / 100.0, bl[:, 1] / 100.0          # us
        t0 = st.min()
        st, en = st - t0, en - t0
        dur = en - st
        span = float(en.max())
        grid = np.linspace(0, span, 41)[:-1]
        running = [int(((st <= x) & (en > x)).sum()) for x in grid]
        last = np.argsort(en)[-5:]
        longest = np.argsort(dur)[-5:]
        rec = dict(step=t, span_us=span, workgroups=int(len(dur)), sum_us=float(dur.sum()),
                   mean_running=float(dur.sum() / span), dur_us=dict(mean=float(dur.mean()), p50=float(np.median(dur)),
                   p90=float(np.percentile(dur, 90)), p99=float(np.percentile(dur, 99)), max=float(dur.max())),
                   last_start_us=float(st.max()), running_over_time=running,
                   last_to_finish=[dict(wg=int(b), start=float(st[b]), dur=float(dur[b])) for b in last],
                   longest=[dict(wg=int(b), start=float(st[b]), dur=float(dur[b])) for b in longest],
                   start_us_p50_p90_p99=[float(np.percentile(st, q)) for q in (50, 90, 99)])
        out.append(rec)
print(json.dumps(out, indent=1))
