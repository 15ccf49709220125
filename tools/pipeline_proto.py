#!/usr/bin/env python3
"""What a software pipeline of two half batches inside ONE synchronous step is worth (GPU box only):
   stream 0:  send A | retire A
   stream 1:          send B (after send A) | retire B          ... joined before the next step
against the same envs as one batch.  python tools/pipeline_proto.py [n_envs] [steps] [mode ...]
modes: one (one batch), par (both halves start together), pipe (B's send starts when A's send ends)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
K = int(sys.argv[2]) if len(sys.argv) > 2 else 400
modes = sys.argv[3:] or ["one", "par", "pipe"]
W = 20
dev = torch.device("cuda:0")
for mode in modes:
    G = 1 if mode == "one" else 2
    n = N // G
    main = torch.cuda.current_stream(dev)
    streams = [main] + [torch.cuda.Stream(device=dev) for _ in range(G - 1)]
    envs, acts = [], []
    for g in range(G):
        envs.append(pcc_rl_amd.BatchedNetworkEnv(n, device=dev, seed=0, env_gid_base=g * n))
        gen = torch.Generator(device=dev).manual_seed(1234 + g)
        acts.append(torch.rand((64, n), generator=gen, device=dev) * 2 - 1)
        envs[g].reset()
    torch.cuda.synchronize()
    ev_fork, ev_sendA, ev_joinB = torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Event()

    def step(t):
        if G == 1:
            envs[0].step(acts[0][t % 64])
            return
        ev_fork.record(main)
        envs[0].step_send(acts[0][t % 64])
        if mode == "pipe":
            ev_sendA.record(main)
        with torch.cuda.stream(streams[1]):
            streams[1].wait_event(ev_sendA if mode == "pipe" else ev_fork)
            envs[1].step_send(acts[1][t % 64])
        envs[0].step_retire()
        with torch.cuda.stream(streams[1]):
            envs[1].step_retire()
            ev_joinB.record(streams[1])
        main.wait_event(ev_joinB)

    for t in range(W):
        step(t)
    torch.cuda.synchronize()
    marks = []
    c0 = time.perf_counter()
    for q in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main)
        for t in range(W + q * K // 4, W + (q + 1) * K // 4):
            step(t)
        e1.record(main)
        marks.append((e0, e1))
    torch.cuda.synchronize()
    el = time.perf_counter() - c0
    for e in envs:
        e.check_flags()
        e.close()
    print(json.dumps({"mode": mode, "n_envs": N, "steps": K, "ms_per_step": 1e3 * el / K, "env_steps_per_s": N * K / el,
                      "ms_per_step_by_quarter": [a.elapsed_time(b) / (K // 4) for a, b in marks]}), flush=True)
