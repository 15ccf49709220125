#!/usr/bin/env python3
"""Throughput record of the reference's dormant engine options (exactness is their point, but nobody should have to guess
whether a step takes 1 ms or 1 s): USE_LATENCY_NOISE and USE_CWND on two senders run in the event-loop build (one lane per
env runs the reference's event loop over a heap in global memory), USE_CWND on one sender in the lane-serial send loop.
usage: engine_throughput.py [n_envs] [steps]   -> one JSON object"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dev = torch.device("cuda:0")
out = {"n_envs": N, "steps": K, "note": "ms per step and env-steps/s over steps 20..20+K of an episode, U(-1,1) actions, default link ranges"}
for name, kw in (("plain", {}), ("use_cwnd", {"use_cwnd": True}), ("latency_noise_1.1", {"latency_noise": 1.1}),
                 ("cwnd_and_noise", {"use_cwnd": True, "latency_noise": 1.1}), ("two_senders_plain", {"n_senders": 2}),
                 ("two_senders_cwnd", {"n_senders": 2, "use_cwnd": True}), ("two_senders_noise", {"n_senders": 2, "latency_noise": 1.1})):
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, ring_capacity=8192, **kw)
    S, A = env.n_senders, env.action_dim
    gen = torch.Generator(device=dev).manual_seed(5)
    acts = torch.rand((20 + K, N, S * A), generator=gen, device=dev) * 2 - 1
    env.reset()
    for t in range(20):
        env.step(acts[t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(20, 20 + K):
        env.step(acts[t])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    env.check_flags()
    out[name] = {"ms_per_step": round(1e3 * el / K, 4), "env_steps_per_s": round(N * K / el)}
    env.close()
    del env
    torch.cuda.empty_cache()
print(json.dumps(out))
