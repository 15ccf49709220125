#!/usr/bin/env python3
"""A few steps of one send_scaling case, for rocprofv3 runs (GPU box only).
usage: light_only.py light|heavy|bench waves_per_cu steps"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd
dev = torch.device("cuda:0")
case, waves, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
N = 65536
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
if case == "light":
    env.randomize_link_params(((400.0, 0.3, 0.0, 0.0, 0.5), (500.0, 0.5, 8.0, 0.05, 0.9)))
    env.set_tuning(heavy_predict=1e18, send_waves=waves, takeover_lanes=0)
elif case == "heavy":
    env.randomize_link_params(((100.0, 0.05, 6.5, 0.0, 1.6), (400.0, 0.1, 8.0, 0.02, 2.4)))
    env.set_tuning(heavy_predict=0.0, send_waves=waves, takeover_lanes=64)
else:
    env.set_tuning(send_waves=waves, heavy_predict=float(os.environ.get("HP", 512)))
env.reset()
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((64, N), generator=gen, device=dev) * 2 - 1 if case == "bench" else torch.zeros((64, N), device=dev)
for t in range(K):
    env.step(acts[t % 64])
torch.cuda.synchronize()
