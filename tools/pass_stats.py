#!/usr/bin/env python3
"""Counters of the send half's wave passes on the bench workload (GPU box only).
usage: PCC_DEBUG_TIMELINE=1 pass_stats.py '[{knobs}, ...]' [n_envs] [steps] [senders]
With two senders the slots read: pass_empty/pk_empty = chain passes settled side by side, pass_scan/pk_scan = token passes,
pass_free/pk_free = 256-packet sweep passes, pass_serial/pk_serial = plain recurrence, scan_nothing_committed = sweeps of
both kinds, cycles_committed = token passes' cycles, cycles_serial = the other passes' cycles."""
import json, os, sys
os.environ.setdefault("PCC_DEBUG_TIMELINE", "2")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd

knob_sets = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [{}]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 150
NS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((K, N, NS) if NS > 1 else (K, N), generator=gen, device=dev) * 2 - 1
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, n_senders=NS)
for knobs in knob_sets:
    env.set_tuning(**knobs)
    env.reset()
    env.debug_pass_stats(reset=True)
    sent0 = int(env.state("total_sent").sum().item())
    for t in range(K):
        env.step(acts[t])
    st = env.debug_pass_stats(reset=True)
    st["packets_total"] = int(env.state("total_sent").sum().item()) - sent0
    for a, b in (("pk_scan", "pass_scan"), ("pk_free", "pass_free"), ("pk_empty", "pass_empty"), ("pk_serial", "pass_serial")):
        st["avg_" + a] = st[a] / max(1, st[b])
    print(json.dumps({"knobs": knobs, "stats": st}), flush=True)
