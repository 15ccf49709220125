#!/usr/bin/env python3
"""First (env, step) where the GPU and the oracle disagree on a Philox batch, with the env's link (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import oracle, pcc_rl_amd
n_envs, n_steps, seed = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
knobs = eval(sys.argv[4]) if len(sys.argv) > 4 else {}
env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device="cuda:0", seed=seed, record_steps=True, auto_reset=False)
env.set_tuning(**knobs)
env.reset()
rs = np.random.RandomState(seed)
acts = rs.uniform(-1, 1, (n_envs, n_steps))
a = torch.as_tensor(acts, dtype=torch.float64, device="cuda:0")
rows = []
for t in range(n_steps):
    o, r, d, info = env.step(a[:, t]); rows.append(info["steps"].clone())
got = torch.stack(rows, 1).cpu().numpy()
ref = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=seed, want_obs=False)
bad = np.argwhere((got != ref["steps"]).any(-1))
print("mismatching (env, step) pairs:", len(bad))
seen = set()
for e, t in bad:
    if e in seen: continue
    seen.add(e)
    p = ref["params"][e]
    print("env", e, "first bad step", t, "params bw,dl,queue,loss,rate0", [repr(float(x)) for x in p], "maxq", p[2] / p[0], "ebw", 1 / p[0])
    print("   gpu", got[e, t, :7].tolist()); print("   ref", ref["steps"][e, t, :7].tolist())
    if t: print("   prev", ref["steps"][e, t - 1, :7].tolist())
    if len(seen) >= 6: break
