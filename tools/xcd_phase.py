#!/usr/bin/env python3
"""Which XCD do the first eight workgroups of the send launch and of the retire launch run on, step by step -- and does a handle
whose retire launch is slow have the two launches out of phase?  (GPU box, profile build.)  python tools/xcd_phase.py [handles]"""
import json, os, sys
os.environ.setdefault("PCC_DEBUG_TIMELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import pcc_rl_amd
N = 65536
dev = torch.device("cuda:0")
out = []
for h in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
    gen = torch.Generator(device=dev).manual_seed(1234)
    acts = torch.rand((400, N, 1), generator=gen, device=dev) * 2 - 1
    env.reset()
    rb = (N + 7) // 8 + 1
    phases, ts, tr = [], [], []
    for t in range(120):
        e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        e[0].record(); env.step_send(acts[t]); e[1].record(); env.step_retire(); e[2].record()
        if t >= 20 and t % 10 == 0:
            torch.cuda.synchronize()
            raw = env.debug_timeline().astype(np.int64)
            reg = raw[2 * N + 2 * rb:2 * N + 2 * rb + 4].reshape(-1)
            phases.append(([int(v & 0xFF) for v in reg[16:24]], [int(v & 0xFF) for v in reg[24:32]]))
        if t >= 20:
            torch.cuda.synchronize()
            ts.append(e[0].elapsed_time(e[1])); tr.append(e[1].elapsed_time(e[2]))
    out.append({"handle": h, "send_ms": round(sum(ts) / len(ts), 4), "retire_ms": round(sum(tr) / len(tr), 4),
                "xcd_of_blocks_0_7_send_then_retire": phases[:3], "all_steps_alike": all(p == phases[0] for p in phases)})
    env.close()
print(json.dumps(out, indent=1))
