#!/usr/bin/env python3
"""How the two send paths scale with the number of persistent wavefronts per compute unit
(GPU box only): every env gets the same fixed link, so all items are alike.
 light: under-driven link, ~P packets per env and MI, lane rounds only
 heavy: saturated deep queue, wave passes only"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd
dev = torch.device("cuda:0")
N = 65536
# per-env random links inside narrow ranges (identical links would put every env's ring writes on the same
# memory channels: an artefact): (bw, lat, queue exponent, loss, rate0 / bw) lo / hi
cases = [("light ~200pk under-driven", ((400.0, 0.3, 0.0, 0.0, 0.5), (500.0, 0.5, 8.0, 0.05, 0.9)), 1e18),
         ("heavy saturated deep queues", ((100.0, 0.05, 6.5, 0.0, 1.6), (400.0, 0.1, 8.0, 0.02, 2.4)), 0.0)]
sel = os.environ.get('PCC_SCALING_CASES')
wl = [int(x) for x in os.environ.get('PCC_SCALING_WAVES', '1,2,4,8,16,32').split(',')]
for name, params, hp in cases:
    if sel and not name.startswith(sel):
        continue
    for waves in wl:
        env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, record_steps=True)
        env.randomize_link_params(params)
        env.set_tuning(heavy_predict=hp, send_waves=waves, takeover_lanes=0 if hp > 1 else 64)
        env.reset()
        zero = torch.zeros((N,), device=dev)
        for t in range(45):
            env.step(zero)
        torch.cuda.synchronize()
        tot_ms, pk = 0.0, 0.0
        K = 8
        for t in range(K):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); env.step_send(zero); e1.record()
            o, r, d, info = env.step_retire()
            torch.cuda.synchronize()
            tot_ms += e0.elapsed_time(e1)
            pk += float(info["steps"][:, 0].mean().item())
        pk /= K; ms = tot_ms / K
        n_waves = 256 * waves
        if hp > 1:   # light: chunks of 64 envs, pk iterations each
            per_wave_chunks = (N / 64) / n_waves
            print(json.dumps({"case": name, "waves_per_cu": waves, "packets_per_env": pk, "send_ms": ms,
                              "ns_per_iteration_per_wave": 1e6 * ms / (pk * max(1.0, per_wave_chunks)),
                              "packets_per_ns_chip": N * pk / (ms * 1e6)}), flush=True)
        else:
            per_wave_envs = N / n_waves
            print(json.dumps({"case": name, "waves_per_cu": waves, "packets_per_env": pk, "send_ms": ms,
                              "ns_per_packet_per_wave": 1e6 * ms / (pk * per_wave_envs),
                              "packets_per_ns_chip": N * pk / (ms * 1e6)}), flush=True)
        env.close()
