#!/usr/bin/env python3
"""Sweep of the send tuning knobs on the bench workload with a FRESH handle per knob set (an old handle's episodes are
slower than a new one's: see `episodes` mode), send and retire timed apart with HIP events (GPU box only).
usage: sweep3.py '[{"heavy_predict": 512}, ...]' [n_envs] [steps] [episodes_per_handle]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd

knob_sets = json.loads(sys.argv[1]) if len(sys.argv) > 1 else [{}]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
K = int(sys.argv[3]) if len(sys.argv) > 3 else 400
EPS = int(sys.argv[4]) if len(sys.argv) > 4 else 1
AUTO = len(sys.argv) > 5 and sys.argv[5] == "auto"   # later episodes start by the env's own auto-reset, not by reset()
W = 20
dev = torch.device("cuda:0")
gen = torch.Generator(device=dev).manual_seed(1234)
POOL = 400   # one action vector per episode step (a short cycled pool makes every env's rate drift: see bench.py)
for knobs in knob_sets:
    kw = {k[4:]: v for k, v in knobs.items() if k.startswith("env_")}
    gen.manual_seed(1234)
    acts = torch.rand((POOL, N, kw.get("n_senders", 1)), generator=gen, device=dev) * 2 - 1
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, **kw)
    env.set_tuning(**{k: v for k, v in knobs.items() if not k.startswith("env_")})
    for ep in range(EPS):
        if ep == 0 or not AUTO:
            env.reset()
            for t in range(W):
                env.step(acts[t % POOL])
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            ev[k][0].record(); env.step_send(acts[(W + k) % POOL]); ev[k][1].record(); env.step_retire(); ev[k][2].record()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        send = [e[0].elapsed_time(e[1]) for e in ev]
        ret = [e[1].elapsed_time(e[2]) for e in ev[:-1]]
        env.check_flags()
        q = lambda x: [round(sum(x[i * len(x) // 4:(i + 1) * len(x) // 4]) / (len(x) // 4), 4) for i in range(4)]
        print(json.dumps({"knobs": knobs, "episode": ep, "ms_per_step": round(1e3 * el / K, 4), "send_ms": round(sum(send) / K, 4),
                          "retire_ms": round(sum(ret) / len(ret), 4), "send_q": q(send), "retire_q": q(ret),
                          "send_ms_max": round(max(send), 4)}), flush=True)
    env.close()
    del env
    torch.cuda.empty_cache()
