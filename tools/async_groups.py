#!/usr/bin/env python3
"""65 536 envs as G independent groups, each a handle on its own stream, free-running (no per-step
sync between the groups): what double-buffered sampling (policy on one group while the other
simulates) can reach.  GPU box only.   python tools/async_groups.py [G ...]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd
N, K, W = 65536, 800, 50
dev = torch.device("cuda:0")
out = []
for G in [int(x) for x in sys.argv[1:]] or [1, 2, 4]:
    n = N // G
    streams = [torch.cuda.Stream(device=dev) for _ in range(G)]
    envs, acts = [], []
    for g in range(G):
        with torch.cuda.stream(streams[g]):
            envs.append(pcc_rl_amd.BatchedNetworkEnv(n, device=dev, seed=0, env_gid_base=g * n))
            for k, v in os.environ.items():   # PCC_TUNE_SEND_WAVES=6 ... : set_tuning(send_waves=6) on every group
                if k.startswith("PCC_TUNE_"):
                    envs[g].set_tuning(**{k[9:].lower(): float(v)})
            gen = torch.Generator(device=dev).manual_seed(1234 + g)
            acts.append(torch.rand((400, n), generator=gen, device=dev) * 2 - 1)   # one action vector per episode step, like bench.py
            envs[g].reset()
    torch.cuda.synchronize()
    def run(t0, t1):
        for t in range(t0, t1):
            for g in range(G):
                with torch.cuda.stream(streams[g]):
                    envs[g].step(acts[g][t % 400])
    run(0, W)
    torch.cuda.synchronize(); c0 = time.perf_counter()
    run(W, W + K)
    torch.cuda.synchronize(); el = time.perf_counter() - c0
    for e in envs:
        e.check_flags(); e.close()
    out.append({"groups": G, "envs_per_group": n, "env_steps_per_s": N * K / el, "ms_per_step_all_groups": 1e3 * el / K})
    print(out[-1], flush=True)
print(json.dumps(out))
