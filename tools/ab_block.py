#!/usr/bin/env python3
"""A/B of run-time tuning knobs on ONE handle in BLOCKS of steps whose order rotates from episode to episode: every setting
sees every stretch of the episode equally often (tools/ab_knob.py does not: its first setting always looks best), and the
first steps after a switch are left out (tools/ab_step.py switches every step: settings that move envs between the light and
the wave path disturb each other -- heavy_predict 384 measured 0.106 ms next to 448 and 0.092 next to itself).
usage: ab_block.py '[{"heavy_predict": 384}, {"heavy_predict": 480}]' [n_envs] [rounds] [n_senders] [block]
       (one round = len(settings) episodes)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd

sets = json.loads(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
ROUNDS = int(sys.argv[3]) if len(sys.argv) > 3 else 1
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
B = int(sys.argv[5]) if len(sys.argv) > 5 else 40
SKIP = 8
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, n_senders=S)
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N, S), generator=gen, device=dev) * 2 - 1
env.reset()
K = len(sets)
EPS = K * ROUNDS
n = 400 * EPS
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(n)]
which, pos = [], []
cur = -1
for k in range(n):
    e, t = divmod(k, 400)
    j = (t // B + e) % K          # the block's setting: rotated by one from episode to episode
    if j != cur:
        env.set_tuning(**sets[j]); cur = j
    which.append(j); pos.append(t % B)
    ev[k][0].record(); env.step_send(acts[t]); ev[k][1].record(); env.step_retire(); ev[k][2].record()
torch.cuda.synchronize()
env.check_flags()
for j, s in enumerate(sets):
    ks = [k for k in range(n) if which[k] == j and pos[k] >= SKIP and (k + 1) % 400 != 0 and k % 400 != 0]
    send = sum(ev[k][0].elapsed_time(ev[k][1]) for k in ks) / len(ks)
    ret = sum(ev[k][1].elapsed_time(ev[k][2]) for k in ks) / len(ks)
    print(json.dumps({"knobs": s, "steps": len(ks), "send_ms": round(send, 4), "retire_ms": round(ret, 4), "sum_ms": round(send + ret, 4)}))
