#!/usr/bin/env python3
"""Cost per packet of the send kernel's paths (diagnostics; GPU box only): every env gets the same
fixed link so all lanes are in one regime; reports ns of send-kernel time per packet of one env."""
import os, sys, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcc_rl_amd

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
res = []
for name, params in [("drop-dominated (bw 105, rate 1000, queue 2981)", (105.0, 0.05, 2981.0, 0.0, 1000.0)),
                     ("middle (bw 350, rate 600, queue 3000)", (350.0, 0.05, 3000.0, 0.01, 600.0)),
                     ("accept-dominated (bw 500, rate 400, dl 0.5)", (500.0, 0.5, 3000.0, 0.01, 400.0))]:
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, link_params=params, record_steps=True)
    if os.environ.get("PCC_HEAVY_PACKETS"):
        env.set_tuning(heavy_packets=float(os.environ["PCC_HEAVY_PACKETS"]), heavy_rho=float(os.environ.get("PCC_HEAVY_RHO", 0.45)))
    env.reset()
    zero = torch.zeros((N,), device=dev)
    for t in range(40):
        env.step(zero)
    torch.cuda.synchronize()
    tot_ms, tot_pk = 0.0, 0.0
    for t in range(10):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.step_send(zero)
        e1.record()
        o, r, d, info = env.step_retire()
        torch.cuda.synchronize()
        tot_ms += e0.elapsed_time(e1)
        tot_pk += float(info["steps"][0, 0])
    lanes_per_wave = min(N, 64)
    res.append({"regime": name, "n_envs": N, "packets_per_env_step": tot_pk / 10, "send_ms": tot_ms / 10,
                "ns_per_packet_of_one_env": 1e6 * tot_ms / tot_pk,
                "ns_per_packet_if_lanes_serialised": 1e6 * tot_ms / tot_pk / lanes_per_wave})
    env.close()
print(json.dumps(res, indent=1))
