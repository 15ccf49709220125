#!/usr/bin/env python3
"""Randomized parity soak on the GPU box (not collected by pytest: run it by hand).

    python tests/soak_parity.py [cases] [first_seed] [senders: 1, 2 or 0 = both]
    (PCC_SOAK_LISTS=1: small batches step by send_kernel + retire_kernel with work lists too; PCC_SOAK_BIG=1: 8 192+ envs only)

Every case: a random batch size, episode stretch, action range (drifting rates up, down or both) and random speed knobs that
move envs between the lane rounds, the wave path's passes and the small / full-size launches -- and the whole batch against the
oracle (test infrastructure: oracle/), every step column and observation bit for bit.  Prints one line per case and stops at
the first difference with what it takes to reproduce it."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import oracle
import pcc_rl_amd

CASES = int(sys.argv[1]) if len(sys.argv) > 1 else 20
SEED0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
SENDERS = int(sys.argv[3]) if len(sys.argv) > 3 else 0
DEV = "cuda:0"
BIG = [8192, 9000, 12288, 16384]
if os.environ.get("PCC_SOAK_BIG"):   # only batches that take the two-launch step
    BIG = [8192, 9000, 12288, 16384]
if os.environ.get("PCC_SOAK_LISTS"):   # small batches through the work lists too (what tests/conftest.py does for the parity suite)
    pcc_rl_amd.BatchedNetworkEnv.DEFAULT_LIST_MIN_ENVS = 0
bad = 0
for c in range(CASES):
    seed = SEED0 + c
    rs = np.random.RandomState(seed)
    ns = SENDERS if SENDERS else 1 + (c & 1)
    # (batches below 8 192 envs step in one launch -- step_small_kernel; from 8 192 up by send_kernel + retire_kernel with work lists)
    n_envs = int(rs.choice(BIG if (c % 3 == 2 or os.environ.get("PCC_SOAK_BIG")) else [96, 200, 512, 1024, 3000]))
    n_steps = int(rs.choice([40, 70, 110])) if n_envs < 8192 else int(rs.choice([30, 50]))
    lo, hi = [(-1.0, 1.0), (-0.5, 1.8), (-1.0, 2.5), (-2.0, 1.0), (-0.2, 0.6)][int(rs.randint(5))]
    knobs = {}
    if rs.rand() < 0.7:
        knobs["heavy_predict"] = float(rs.choice([60.0, 100.0, 200.0, 480.0, 1e9]))
    if rs.rand() < 0.5:
        knobs["takeover_lanes"] = int(rs.choice([0, 8, 32, 64]))
    if rs.rand() < 0.5:
        knobs["round_packets"] = int(rs.choice([4, 8, 64, 256]))
    if rs.rand() < 0.3:
        knobs["send_waves"] = int(rs.choice([2, 8, 13]))
    if rs.rand() < 0.3:
        knobs["send_envs_per_wave"] = int(rs.choice([16, 40, 64]))
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, n_senders=ns, record_steps=True, auto_reset=False)
    env.set_tuning(**knobs)
    env.reset()
    acts = rs.uniform(lo, hi, (n_envs, n_steps, ns) if ns > 1 else (n_envs, n_steps))
    a = torch.as_tensor(acts, dtype=torch.float64, device=DEV)
    rows, obs = [], []
    for t in range(n_steps):
        o, r, d, info = env.step(a[:, t])
        rows.append(info["steps"].clone()); obs.append(o.clone())
    torch.cuda.synchronize()
    env.check_flags()
    ax = 1 if ns == 1 else 2
    steps = torch.stack(rows, ax).cpu().numpy(); ob = torch.stack(obs, ax).cpu().numpy()
    ref = oracle.run_batch(acts, n_senders=ns, rng_mode=oracle.RNG_PHILOX, seed=seed)
    ok = np.array_equal(steps, ref["steps"]) and np.array_equal(ob, ref["obs"].astype(np.float32))
    pk = float(ref["steps"][..., 0].mean())
    print(json.dumps({"case": c, "seed": seed, "senders": ns, "envs": n_envs, "steps": n_steps, "actions": [lo, hi], "knobs": knobs,
                      "packets_per_sender_step": round(pk, 1), "equal": bool(ok)}), flush=True)
    env.close()
    if not ok:
        bad += 1
        w = np.argwhere(steps != ref["steps"])
        print("FIRST DIFFERENCE at", w[0].tolist() if len(w) else "obs only")
        break
print("soak: %d cases, %s" % (c + 1, "ALL EQUAL" if not bad else "DIFFERENCE FOUND"))
sys.exit(1 if bad else 0)
