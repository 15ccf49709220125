#!/usr/bin/env python3
"""Sweep of the send tuning knobs on the bench workload, send and retire timed apart with HIP events
(GPU box only).  usage: sweep2.py '[{"heavy_predict": 512}, ...]' [n_envs] [steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd

knob_sets = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [{}]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
W = 20
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((64, N), generator=gen, device=dev) * 2 - 1
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
out = []
for knobs in knob_sets:
    env.set_tuning(**knobs)
    env.reset()
    for t in range(W):
        env.step(acts[t % 64])
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record(); env.step_send(acts[(W + k) % 64]); ev[k][1].record(); env.step_retire(); ev[k][2].record()
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    send = [e[0].elapsed_time(e[1]) for e in ev]
    ret = [e[1].elapsed_time(e[2]) for e in ev[:-1]]
    env.check_flags()
    rec = {"knobs": knobs, "ms_per_step": 1e3 * el / K, "send_ms": sum(send) / K, "retire_ms": sum(ret) / len(ret),
           "send_ms_by_quarter": [sum(send[i * K // 4:(i + 1) * K // 4]) / (K // 4) for i in range(4)],
           "send_ms_max": max(send)}
    out.append(rec)
    print(json.dumps(rec), flush=True)
