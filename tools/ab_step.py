#!/usr/bin/env python3
"""A/B of run-time tuning knobs on ONE handle with the settings alternating EVERY step: every setting sees every phase of the
episodes (tools/ab_knob.py's blocks of 25 steps line up with the episode's growth of the packet counts: its first setting
always looks best).  Send and retire launches timed apart with HIP events; the step that runs the boundary reset is left out.
usage: ab_step.py '[{"light_wgs": 40}, {"light_wgs": 0}]' [n_envs] [episodes] [n_senders]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd

sets = json.loads(sys.argv[1])
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
EPS = int(sys.argv[3]) if len(sys.argv) > 3 else 3
S = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, n_senders=S)
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N, S), generator=gen, device=dev) * 2 - 1
env.reset()
K = 400 * EPS
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
for k in range(K):
    env.set_tuning(**sets[k % len(sets)])
    ev[k][0].record(); env.step_send(acts[k % 400]); ev[k][1].record(); env.step_retire(); ev[k][2].record()
torch.cuda.synchronize()
env.check_flags()
for j, s in enumerate(sets):
    ks = [k for k in range(K) if k % len(sets) == j and (k + 1) % 400 != 0 and k % 400 != 0]
    send = sum(ev[k][0].elapsed_time(ev[k][1]) for k in ks) / len(ks)
    ret = sum(ev[k][1].elapsed_time(ev[k][2]) for k in ks) / len(ks)
    print(json.dumps({"knobs": s, "steps": len(ks), "send_ms": round(send, 4), "retire_ms": round(ret, 4), "sum_ms": round(send + ret, 4)}))
