#!/usr/bin/env python3
"""Dump per-env packets sent / accepted fraction at a few steps (diagnostics; GPU box only)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcc_rl_amd
N = 65536
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, record_steps=True)
env.reset()
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((64, N), generator=gen, device=dev) * 2 - 1
out = {}
for t in range(399):
    o, r, d, info = env.step(acts[t % 64])
    if t in (20, 50, 100, 200, 300, 398):
        s = info["steps"]
        out["sent_%d" % t] = s[:, 0].cpu().numpy().astype(np.int32)
        out["lost_%d" % t] = s[:, 2].cpu().numpy().astype(np.int32)
np.savez_compressed("gpurun_out/sent_dump.npz", **out)
print("ok")
