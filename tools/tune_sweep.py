#!/usr/bin/env python3
"""Quick sweep of the send tuning knobs on the bench workload (fused step); GPU box only."""
import itertools, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd
N, K = 65536, 400
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((64, N), generator=gen, device=dev) * 2 - 1
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
def run(**knobs):
    base = dict(takeover_lanes=1, help_lanes=16, heavy_predict=3072.0, round_packets=256, send_waves=4)
    base.update(knobs)
    env.set_tuning(**base)
    env.reset()
    for t in range(30):
        env.step(acts[t % 64])
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for t in range(30, 30 + K):
        env.step(acts[t % 64])
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
res = []
for knobs in [dict(), dict(send_envs_per_wave=48), dict(send_envs_per_wave=32), dict(send_envs_per_wave=32, heavy_predict=4096.0), dict(send_envs_per_wave=32, send_waves=2, heavy_predict=4096.0), dict(send_envs_per_wave=64, help_lanes=24)]:
    ms = run(**knobs)
    res.append((knobs, ms))
    print(knobs, "%.4f ms/step" % ms, flush=True)
