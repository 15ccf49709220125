#!/usr/bin/env python3
"""Send and retire launch times step by step over some episodes (GPU box): which steps of an episode are the slow ones?
   python tools/step_times.py [n_envs] [senders] [episodes]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pcc_rl_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
EPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, n_senders=S)
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N, S), generator=gen, device=dev) * 2 - 1
env.reset()
n = 400 * EPS
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
for k in range(n):
    ev[k][0].record(); env.step_send(acts[k % 400]); ev[k][1].record(); env.step_retire(); ev[k][2].record()
torch.cuda.synchronize()
env.check_flags()
send = np.array([e[0].elapsed_time(e[1]) for e in ev]) * 1e3
ret = np.array([e[1].elapsed_time(e[2]) for e in ev]) * 1e3
for name, v in (("send_us", send), ("retire_us", ret)):
    by_step = v.reshape(EPS, 400)
    print(json.dumps({"launch": name, "mean": round(float(v.mean()), 1), "p10_p50_p90_p99_max": [round(float(x), 1) for x in np.percentile(v, [10, 50, 90, 99, 100])],
                      "mean_by_stretch_of_the_episode": {"0-4": round(float(by_step[:, :5].mean()), 1), "5-19": round(float(by_step[:, 5:20].mean()), 1),
                                                         "20-99": round(float(by_step[:, 20:100].mean()), 1), "100-399": round(float(by_step[:, 100:].mean()), 1)},
                      "slowest_steps": [{"episode": int(k // 400), "step": int(k % 400), "us": round(float(v[k]), 1)} for k in np.argsort(-v)[:12]],
                      "us_above_the_median_summed": round(float(np.clip(v - np.median(v), 0, None).sum()), 0), "total_us": round(float(v.sum()), 0)}), flush=True)
