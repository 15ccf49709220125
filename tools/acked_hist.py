#!/usr/bin/env python3
"""Histogram of acknowledged packets per env and MI (the lengths of the RTT lists retire sums); GPU box only."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcc_rl_amd
from pcc_rl_amd import native
N = 65536
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, record_steps=True)
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((64, N), generator=gen, device=dev) * 2 - 1
env.reset()
col = native.STEP_COLUMNS.index("acked") if "acked" in native.STEP_COLUMNS else 1
out = []
for t in range(400):
    o, r, d, info = env.step(acts[t % 64])
    if t in (10, 50, 100, 200, 300, 398):
        a = info["steps"][:, col].cpu().numpy()
        edges = [0, 1, 9, 65, 129, 257, 513, 1025, 2049, 4097, 1 << 30]
        h, _ = np.histogram(a, bins=edges)
        wave_max = a.reshape(-1, 4).max(axis=1)
        item_max = a.reshape(-1, 16).max(axis=1)
        leaves = np.where(a <= 128, 1, np.ceil(a / 100.0))
        out.append({"step": t, "mean": float(a.mean()), "max": float(a.max()), "edges": edges[:-1], "hist": h.tolist(),
                    "mean_of_max_over_4": float(wave_max.mean()), "mean_of_max_over_16": float(item_max.mean()),
                    "frac_gt_128": float((a > 128).mean()), "frac_gt_256": float((a > 256).mean()),
                    "mean_leaves": float(leaves.mean()), "mean_max4_leaves": float(leaves.reshape(-1, 4).max(axis=1).mean())})
print(json.dumps(out))
