#!/usr/bin/env python3
"""Throughput of the other BASELINE.json configs (diagnostics; GPU box only; not the bench line):
config 2 (4 096 envs, fixed link) and config 5 (32 768 envs x 2 senders)."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcc_rl_amd

dev = torch.device("cuda:0")
out = []
for name, kw in [("config2: 4096 envs, fixed link (200 pkt/s, 30 ms, queue 5, no loss), rate0 60",
                  dict(n_envs=4096, link_params=(200.0, 0.03, 5.0, 0.0, 60.0))),
                 ("config5: 32768 envs x 2 senders on one bottleneck, randomized links",
                  dict(n_envs=32768, n_senders=2)),
                 ("config3 at 4096 envs (randomized links)", dict(n_envs=4096))]:
    env = pcc_rl_amd.BatchedNetworkEnv(device=dev, seed=0, **kw)
    N, S = env.n_envs, env.n_senders
    gen = torch.Generator(device=dev).manual_seed(0)
    acts = torch.rand((64, N, S), generator=gen, device=dev) * 2 - 1
    env.reset()
    for t in range(50):
        env.step(acts[t % 64])
    sent0 = env.state("total_sent").sum()
    torch.cuda.synchronize()
    K = 800
    t0 = time.perf_counter()
    for t in range(K):
        env.step(acts[t % 64])
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    env.check_flags()
    pk = float((env.state("total_sent").sum() - sent0).item()) / (N * K)
    out.append({"config": name, "env_steps_per_s": N * K / dt, "ms_per_step": 1e3 * dt / K, "packets_per_env_step": pk})
    env.close()
print(json.dumps(out, indent=1))
json.dump(out, open("gpurun_out/other_configs.json", "w"), indent=1)
