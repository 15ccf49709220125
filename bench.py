#!/usr/bin/env python3
"""Benchmark of the hot path: env steps/s of the batched simulator on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one the metric is quoted on): 65 536 parallel envs
per GPU, single sender, per-env randomized bandwidth / latency / queue / loss over the ICML'19
ranges, actions U(-1, 1) pre-generated on the device, Philox loss uniforms, auto-reset at the
400-step episode boundary.  One "step" = one `step()` of all envs = one monitor interval per
env.  For N > 1 every rank owns its own 65 536 envs (weak scaling, env ids rank*65536+i, no
collective inside the step; episode returns are all-gathered over RCCL when episodes end).

Prints ONE JSON line on rank 0 (see README / DESIGN.md section 6 for the fields).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import pcc_rl_amd  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
B_FIXED = 450            # algorithmic bytes per env-step excluding packets
B_PACKET = 32            # one 16-B in-flight record written at send + read at completion


def cpu_baseline(seconds_budget=20.0, threads=1):
    """The CPU oracle (oracle/pcc_oracle.c, the literal heap-based restatement of the
    reference engine) timed on this host on a bounded sample of the same workload."""
    import numpy as np

    import oracle

    rs = np.random.RandomState(0)
    n_envs, n_steps = 64, 100
    done_steps, done_pk, t_used = 0, 0.0, 0.0
    base = 0
    while t_used < seconds_budget:
        acts = rs.uniform(-1, 1, (n_envs, n_steps))
        t0 = time.perf_counter()
        out = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=0, env_gid_base=base,
                               n_threads=threads, want_obs=False)
        t_used += time.perf_counter() - t0
        done_steps += n_envs * n_steps
        done_pk += float(out["steps"][..., 0].sum())
        base += n_envs
    return {"value": done_steps / t_used, "unit": "env steps/s", "cores": threads, "kind": "port",
            "sample": "%d env-steps (%d-env x %d-step batches of the bench workload, Philox uniforms, "
                      "%.1f packets/step) on the C oracle, %d thread(s), %.1f s; host has %d cores"
                      % (done_steps, n_envs, n_steps, done_pk / done_steps, threads, t_used, os.cpu_count() or 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    N, K, W = args.envs, args.steps, args.warmup
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, env_gid_base=rank * N, auto_reset=True)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    pool = 64
    actions = torch.rand((pool, N), generator=gen, device=dev, dtype=torch.float32) * 2 - 1
    env.reset()
    returns_gathered = 0
    gather_buf = torch.empty((world * N,), dtype=torch.float32, device=dev) if world > 1 else None

    def one_step(t):
        nonlocal returns_gathered
        env.step(actions[t % pool])
        if world > 1 and (t + 1) % env.max_steps == 0:
            # the only inter-GPU traffic on this path: episode returns, once per episode
            dist.all_gather_into_tensor(gather_buf, env.episode_returns().to(torch.float32))
            returns_gathered += 1

    t_global = 0
    for _ in range(W):
        one_step(t_global)
        t_global += 1
    sent0 = env.state("total_sent").sum()

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        ev[k][0].record()
        one_step(t_global)
        ev[k][1].record()
        t_global += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    packets = float((env.state("total_sent").sum() - sent0).item())
    env.check_flags()
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    # steps that also ran the episode-boundary reset kernel are kept out of the step-kernel average
    first = W
    plain = [kernel_ms[k] for k in range(K) if (first + k + 1) % env.max_steps != 0]
    step_kernel_ms = sum(plain) / max(1, len(plain))

    stats = torch.tensor([elapsed, packets], dtype=torch.float64, device=dev)
    if world > 1:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        elapsed, packets_all = float(mx[0].item()), float(sm[1].item())
    else:
        packets_all = packets

    if rank == 0:
        value = world * N * K / elapsed
        pk_per_step = packets / (N * K)
        alg_bytes_per_launch = N * (B_FIXED + B_PACKET * pk_per_step)
        achieved = alg_bytes_per_launch / (step_kernel_ms * 1e-3) / 1e9
        out = {
            "metric": "env steps/sec (whole node) at 64k parallel envs",
            "value": value, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d envs/GPU, 1 sender, per-env randomized bw/latency/queue/loss "
                                   "(ICML'19 ranges), U(-1,1) actions, 400-step episodes, auto-reset" % N,
                       "envs_per_gpu": N, "packets_per_env_step": pk_per_step,
                       "episode_return_allgathers": returns_gathered},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBPS, "traffic": None,
                         "kernel": "step_kernel<1>", "kernel_ms": step_kernel_ms,
                         "algorithmic_bytes_per_launch": alg_bytes_per_launch},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
