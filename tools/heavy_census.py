#!/usr/bin/env python3
"""Which envs are heavy, and in which regime (diagnostics; GPU box only): for a few steps of the
bench workload, the envs with the most packets per MI with their rate/bw ratio and loss."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcc_rl_amd

N = 65536
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, record_steps=True)
env.reset()
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((64, N), generator=gen, device=dev) * 2 - 1
out = []
for t in range(399):
    o, r, d, info = env.step(acts[t % 64])
    if t in (5, 50, 150, 300, 398):
        s = info["steps"]
        sent, acked, lost = s[:, 0], s[:, 1], s[:, 2]
        rate, bw = env.state("rate")[0], env.state("bw")
        rho = (bw / rate)
        big = sent > 512
        top = torch.argsort(sent, descending=True)[:12]
        rec = {"step": t, "n_big": int(big.sum()), "packets_in_big": float(sent[big].sum() / sent.sum()),
               "big_rho_lt_0.5": int((big & (rho < 0.5)).sum()), "big_rho_0.5_0.9": int((big & (rho >= 0.5) & (rho < 0.9)).sum()),
               "big_rho_ge_0.9": int((big & (rho >= 0.9)).sum()),
               "big_drop_frac_mean": float((lost[big] / (acked[big] + lost[big]).clamp(min=1)).mean()) if big.any() else 0.0,
               "top": [[float(sent[i]), float(rate[i]), float(bw[i]), float(lost[i] / max(1.0, float(acked[i] + lost[i]))),
                        float(env.state("maxq")[i])] for i in top]}
        # per-size buckets of drop fraction
        for lo, hi in [(512, 1024), (1024, 2048), (2048, 4096), (4096, 1 << 30)]:
            m = (sent > lo) & (sent <= hi)
            rec["bucket_%d" % lo] = [int(m.sum()), float((lost[m] / (acked[m] + lost[m]).clamp(min=1)).mean()) if m.any() else 0.0]
        out.append(rec)
json.dump(out, open("gpurun_out/heavy_census.json", "w"), indent=1)
for r in out:
    print({k: v for k, v in r.items() if k != "top"})
    print("  top:", [[round(x, 2) for x in row] for row in r["top"][:5]])
