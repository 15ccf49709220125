#!/usr/bin/env python3
"""The same batch stepped by the fused launch and by two launches (GPU box only): first difference, step by step.
   python tools/fused_vs_two.py [n_envs] [steps] [n_senders]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcc_rl_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
T = int(sys.argv[2]) if len(sys.argv) > 2 else 100
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
pcc_rl_amd.BatchedNetworkEnv.DEFAULT_LIST_MIN_ENVS = 0
envs = []
for fused in (0, 1):
    e = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=3, n_senders=S, record_steps=True, auto_reset=False)
    e.set_tuning(fused=fused)
    for k, v in os.environ.items():
        if k.startswith("PCC_TUNE_"):
            e.set_tuning(**{k[9:].lower(): float(v)})
    e.reset()
    envs.append(e)
gen = torch.Generator(device=dev).manual_seed(5)
bad = 0
for t in range(T):
    a = torch.rand((N, S), generator=gen, device=dev) * 2 - 1
    rows = []
    for e in envs:
        o, r, d, info = e.step(a if S > 1 else a[:, 0])
        rows.append(info["steps"].clone().cpu().numpy().reshape(N, S, -1))
    if not np.array_equal(rows[0], rows[1]):
        diff = np.argwhere(rows[0] != rows[1])
        envs_bad = sorted(set(int(x[0]) for x in diff))
        print("step %d: %d envs differ, e.g. env %d cols %s" % (t, len(envs_bad), envs_bad[0], sorted(set(int(x[2]) for x in diff if x[0] == envs_bad[0]))))
        i = envs_bad[0]
        print("   two  :", rows[0][i, 0, :8])
        print("   fused:", rows[1][i, 0, :8])
        bad += 1
        if bad >= 3:
            break
print("fused steps:", envs[1].fused_steps(), "flags:", int(envs[1].state("flags").max().item()), "bad steps:", bad)
