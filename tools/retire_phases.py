#!/usr/bin/env python3
"""Where the retire half's time goes, by phase (PCC_DEBUG_TIMELINE=1; GPU box only): per-wavefront
time in each phase of retire_env, averaged over all wavefronts, at a few steps of an episode."""
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
names = {3: "state loads", 4: "boundary search", 5: "candidates+repairs", 6: "ending event", 7: "write-back",
         8: "rtt_means", 9: "metrics+batch loads", 10: "history+obs", 11: "outputs"}
out = []
prev = None
for t in range(310):
    env.step(acts[t % 64])
    if t in (19, 99, 199, 299):
        prev = env.debug_timeline().astype(np.int64)[2 * N:].reshape(-1, 16)
    if t in (20, 100, 200, 300):
        raw = env.debug_timeline().astype(np.int64)
        # one row per retire workgroup, behind the 2 n item slots of the send half; the phase sums (and the search counters)
        # accumulate over the launches: this launch's share is the difference to the step before
        bl = raw[2 * N:].reshape(-1, 16) - prev
        waves = (N + 3) // 4                        # 16 lanes per env
        rec = {"step": t, "us_per_wavefront": {v: float(bl[:, k].sum()) / 100.0 / waves for k, v in names.items()}}
        rec["us_per_wavefront"]["total"] = sum(rec["us_per_wavefront"].values())
        # (cumulative since the handle was created) boundary searches, those whose predicted window held the boundary,
        # wavefronts, wavefronts that needed no descent at all
        h = [int(bl[:, k].sum()) for k in (12, 13, 14, 15)]
        rec["search_hints"] = {"searches": h[0], "window_hits": h[1], "hit_rate": h[1] / max(1, h[0]),
                               "wavefronts": h[2], "wavefronts_without_descent": h[3], "wave_rate": h[3] / max(1, h[2])}
        out.append(rec)
print(json.dumps(out, indent=1))
