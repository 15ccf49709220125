#!/usr/bin/env python3
"""Per-step packet statistics of the bench workload (diagnostics; GPU box only): for every
step the mean / max / p99 packets sent per env, so per-dispatch kernel times from
`rocprofv3 --kernel-trace` can be set against the work each launch had."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcc_rl_amd  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 450
    out = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/step_stats.json"
    dev = torch.device("cuda:0")
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, ring_capacity=int(os.environ.get("PCC_RING_CAP", 0)),
                                       balance_every=int(os.environ.get("PCC_BALANCE", 0)))
    if os.environ.get("PCC_HEAVY_PACKETS") or os.environ.get("PCC_HEAVY_RHO"):
        env.set_tuning(heavy_packets=float(os.environ.get("PCC_HEAVY_PACKETS", 1e18)),
                       heavy_rho=float(os.environ.get("PCC_HEAVY_RHO", 0.45)))
    if os.environ.get("PCC_HEAVY_PREDICT"):
        env.set_tuning(heavy_predict=float(os.environ["PCC_HEAVY_PREDICT"]))
    if os.environ.get("PCC_EPW"):
        env.set_tuning(send_envs_per_wave=float(os.environ["PCC_EPW"]))
    if os.environ.get("PCC_ROUND") or os.environ.get("PCC_TAKEOVER"):
        env.set_tuning(round_packets=float(os.environ.get("PCC_ROUND", 256)),
                       takeover_lanes=float(os.environ.get("PCC_TAKEOVER", 2)))
    env.reset()
    gen = torch.Generator(device=dev).manual_seed(1234)
    acts = torch.rand((64, N), generator=gen, device=dev) * 2 - 1
    prev = env.state("total_sent").clone()
    rows = []
    for t in range(T):
        env.step(acts[t % 64])
        cur = env.state("total_sent")
        d = (cur - prev).double()
        prev = cur.clone()
        w = d.view(-1, 64).max(dim=1).values
        rows.append({"step": t, "mean": float(d.mean()), "max": float(d.max()),
                     "p99": float(torch.quantile(d[:100000], 0.99)) if N <= 100000 else 0.0,
                     "wave_max_mean": float(w.mean())})
    json.dump(rows, open(out, "w"))
    print(rows[0], rows[1], rows[100], rows[-1])


if __name__ == "__main__":
    main()
