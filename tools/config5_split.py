#!/usr/bin/env python3
"""send/retire split of BASELINE config 5 (32 768 envs x 2 senders); diagnostics, GPU box only."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcc_rl_amd
dev = torch.device("cuda:0")
N = 32768
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, n_senders=2)
gen = torch.Generator(device=dev).manual_seed(0)
acts = torch.rand((64, N, 2), generator=gen, device=dev) * 2 - 1
env.reset()
res = []
for t in range(400):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record(); env.step_send(acts[t % 64]); e[1].record(); env.step_retire(); e[2].record()
    if t % 50 == 0 or t == 399:
        torch.cuda.synchronize()
        sent = env.state("total_sent")
        res.append((t, e[0].elapsed_time(e[1]), e[1].elapsed_time(e[2])))
print(res)
