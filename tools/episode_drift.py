#!/usr/bin/env python3
"""Send and retire launch times episode by episode on one handle (GPU box): does a batch get slower as episodes go by?
   python tools/episode_drift.py [n_envs] [senders] [episodes]
HIP events around each launch, steps 20..398 of every episode, auto-reset on; also the envs per ring tier at step 200."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
EPS = int(sys.argv[3]) if len(sys.argv) > 3 else 8
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, n_senders=S)
for k, v in os.environ.items():
    if k.startswith("PCC_TUNE_"):
        env.set_tuning(**{k[9:].lower(): float(v)})
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N, S), generator=gen, device=dev) * 2 - 1
env.reset()
for e in range(EPS):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(400)]
    for t in range(400):
        ev[t][0].record(); env.step_send(acts[t]); ev[t][1].record(); env.step_retire(); ev[t][2].record()
    torch.cuda.synchronize()
    ks = range(20, 399)
    out = {"episode": e, "send_ms": round(sum(ev[k][0].elapsed_time(ev[k][1]) for k in ks) / len(ks), 4),
           "retire_ms": round(sum(ev[k][1].elapsed_time(ev[k][2]) for k in ks) / len(ks), 4),
           "first20_send_ms": round(sum(ev[k][0].elapsed_time(ev[k][1]) for k in range(20)) / 20, 4),
           "first20_retire_ms": round(sum(ev[k][1].elapsed_time(ev[k][2]) for k in range(20)) / 20, 4)}
    try:
        tier = env.state("ring_tier")
        out["senders_per_ring_tier"] = torch.bincount(tier.reshape(-1).long(), minlength=4).tolist()
    except Exception as ex:
        out["ring_tier"] = str(ex)[:80]
    print(json.dumps(out), flush=True)
env.check_flags()
