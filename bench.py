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
from pcc_rl_amd import distributed as pdist  # noqa: E402

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Algorithmic bytes per env-step: 450 B of fixed traffic + 32 B per packet
# (one 16-B in-flight record written at send, read at completion).  A step is two launches,
# timed apart with HIP events: send_kernel (dominant) reads the action (4), link parameters (32),
# and reads+writes link state (2x16) and sender rate/next_send/cursors (2x26) = 120 B, and writes
# the 16-B record; retire_kernel owns the remaining 330 B (state, history, obs/reward/done) and
# reads the record.  With --fused the step is ONE launch (step_kernel: a workgroup sends for its
# 64 envs, then retires envs of whichever blocks are done) and carries all 450 + 32 P bytes.
B_FIXED_SEND, B_FIXED_RETIRE, B_PACKET_HALF = 120, 330, 16


def pmc_traffic():
    """HBM bytes per launch from the committed PMC summary (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE,
    separate passes over this same script; see profiles/README.md).  Counters cannot be read from
    inside the process, so the figure is the profile's, labelled with its source."""
    for name in ("r01_v9_pmc_hbm.json", "r01_v8_pmc_hbm.json"):
        path = os.path.join(ROOT, "profiles", name)
        if os.path.exists(path):
            with open(path) as f:
                return json.load(f), "profiles/" + name
    return None, None


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
    out = {"value": done_steps / t_used, "unit": "env steps/s", "cores": threads, "kind": "port",
           "sample": "%d env-steps (%d-env x %d-step batches of the bench workload, Philox uniforms, "
                     "%.1f packets/step) on the C oracle, %d thread(s), %.1f s; host has %d cores"
                     % (done_steps, n_envs, n_steps, done_pk / done_steps, threads, t_used, os.cpu_count() or 1)}
    # the same C oracle on every host core (envs are independent: one env batch per thread)
    cores = os.cpu_count() or 1
    if cores > 1:
        n_all, steps_all, t_all, base_all = 8 * cores, 0, 0.0, 1 << 20
        while t_all < 5.0:
            acts = rs.uniform(-1, 1, (n_all, n_steps))
            t0 = time.perf_counter()
            oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=0, env_gid_base=base_all, n_threads=cores,
                             want_obs=False)
            t_all += time.perf_counter() - t0
            steps_all += n_all * n_steps
            base_all += n_all
        out["all_cores"] = {"value": steps_all / t_all, "unit": "env steps/s", "cores": cores,
                            "sample": "%d env-steps on %d threads, %.1f s" % (steps_all, cores, t_all)}
    # the same algorithm in the reference's own language (heapq + numpy), for scale: a few episodes
    from oracle.pcc_oracle_py import time_episodes
    py_steps, py_pk, py_s = time_episodes(1000, 2, n_steps=200)
    out["python_port"] = {"value": py_steps / py_s, "unit": "env steps/s", "cores": 1,
                          "sample": "%d env-steps, %.1f packets/step, %.1f s (oracle/pcc_oracle_py.py)"
                                    % (py_steps, py_pk / py_steps, py_s)}
    return out


def async_groups(N, dev, K, W, n_groups=4):
    """Supplementary figure, NOT the headline: the same N envs as independent groups on their own
    streams, stepped without a per-step synchronization between the groups (double-buffered
    sampling: the policy works on one group while the others simulate).  Same envs, same results,
    same work; the bulk of one group's step fills the send tail of another's."""
    K = min(K, 400)
    env = pcc_rl_amd.GroupedNetworkEnv(N, n_groups, device=dev, seed=0)
    acts = []
    for g in range(n_groups):
        gen = torch.Generator(device=dev).manual_seed(4321 + g)
        acts.append(torch.rand((64, env.group_size), generator=gen, device=dev, dtype=torch.float32) * 2 - 1)
    env.reset()
    for t in range(W):
        for g in range(n_groups):
            env.step_group(g, acts[g][t % 64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        for g in range(n_groups):
            env.step_group(g, acts[g][t % 64])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    env.check_flags()
    env.close()
    return {"value": N * K / el, "unit": "env steps/s", "groups": n_groups, "steps": K,
            "note": "same %d envs as %d independent groups on their own streams, no per-step sync between groups; "
                    "supplementary, the headline value is one synchronous batch" % (N, n_groups)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--envs", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--ring-capacity", type=int, default=0, help="records per accepted ring (0 = library default)")
    ap.add_argument("--groups", type=int, default=0,
                    help="also measure the same envs as this many independent groups on their own streams "
                         "(supplementary field async_groups; how well the groups overlap depends on how HIP maps "
                         "the streams to hardware queues)")
    ap.add_argument("--share-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (lets the N > 1 path run on a 1-GPU box with gloo)")
    args = ap.parse_args()

    rank, world, local_rank = pdist.rank_info()
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        pdist.init_process_group(args.backend, device=dev)   # "nccl" is RCCL on ROCm

    N, K, W = args.envs, args.steps, args.warmup
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, env_gid_base=pdist.env_gid_base(rank, N),
                                       auto_reset=True, ring_capacity=args.ring_capacity)
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    pool = 64
    actions = torch.rand((pool, N), generator=gen, device=dev, dtype=torch.float32) * 2 - 1
    args.split = True
    env.reset()
    returns_gathered = 0
    gather_buf = torch.empty((world * N,), dtype=torch.float32, device=dev) if world > 1 else None

    def one_step(t, ev=None):
        nonlocal returns_gathered
        if ev is not None:
            ev[0].record()
        if args.split:
            env.step_send(actions[t % pool])
            if ev is not None:
                ev[1].record()
            env.step_retire()
        else:
            env.step(actions[t % pool])
        if ev is not None:
            ev[2].record()
        if world > 1 and (t + 1) % env.max_steps == 0:
            # the only inter-GPU traffic on this path: episode returns, once per episode
            pdist.gather_episode_returns(env.episode_returns().to(torch.float32), out=gather_buf)
            returns_gathered += 1

    t_global = 0
    for _ in range(W):
        one_step(t_global)
        t_global += 1
    sent0 = env.state("total_sent").sum()

    # HIP events on the launch stream (torch's current stream IS the stream the library launches on)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(K)]
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(K):
        one_step(t_global, ev[k])
        t_global += 1
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0

    packets = float((env.state("total_sent").sum() - sent0).item())
    if not os.environ.get("PCC_BENCH_IGNORE_FLAGS"):   # experiments only: an overflowed ring means invalid results
        env.check_flags()
    plain = [k for k in range(K) if (W + k + 1) % env.max_steps != 0]
    # steps that also ran the episode-boundary reset kernels are kept out of the kernel averages
    if args.split:
        send_ms = sum(ev[k][0].elapsed_time(ev[k][1]) for k in range(K)) / K
        retire_ms = sum(ev[k][1].elapsed_time(ev[k][2]) for k in plain) / max(1, len(plain))
    else:
        step_ms = sum(ev[k][0].elapsed_time(ev[k][2]) for k in plain) / max(1, len(plain))

    elapsed = pdist.max_over_ranks(elapsed, device=dev)     # MAX over ranks (bench contract)
    max_steps = env.max_steps
    env.close()

    if rank == 0:
        value = world * N * K / elapsed
        pk_per_step = packets / (N * K)
        send_bytes = N * (B_FIXED_SEND + B_PACKET_HALF * pk_per_step)
        retire_bytes = N * (B_FIXED_RETIRE + B_PACKET_HALF * pk_per_step)
        out = {
            "metric": "env steps/sec (whole node) at 64k parallel envs",
            "value": value, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": 1e3 * elapsed / K, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%d envs/GPU, 1 sender, per-env randomized bw/latency/queue/loss "
                                   "(ICML'19 ranges), U(-1,1) actions, 400-step episodes, auto-reset" % N,
                       "envs_per_gpu": N, "packets_per_env_step": pk_per_step,
                       "episode_return_allgathers": returns_gathered},
        }
        if args.split:
            send_gbps = send_bytes / (send_ms * 1e-3) / 1e9
            retire_gbps = retire_bytes / (retire_ms * 1e-3) / 1e9
            both = (send_bytes + retire_bytes) / ((send_ms + retire_ms) * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "send_kernel<1, false>", "achieved": send_gbps,
                               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": send_gbps / HBM_PEAK_GBPS,
                               "traffic": None, "kernel_ms": send_ms, "algorithmic_bytes_per_launch": send_bytes,
                               "other_kernels": [{"kernel": "retire_kernel<1>", "achieved": retire_gbps,
                                                  "frac": retire_gbps / HBM_PEAK_GBPS, "kernel_ms": retire_ms,
                                                  "algorithmic_bytes_per_launch": retire_bytes}],
                               "whole_step": {"achieved": both, "frac": both / HBM_PEAK_GBPS}}
        else:
            # the step IS the dominant kernel: one launch per step
            step_bytes = send_bytes + retire_bytes
            gbps = step_bytes / (step_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "step_kernel<1, false>", "achieved": gbps,
                               "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": gbps / HBM_PEAK_GBPS,
                               "traffic": None, "kernel_ms": step_ms, "algorithmic_bytes_per_launch": step_bytes,
                               "algorithmic_bytes_per_env_step": B_FIXED_SEND + B_FIXED_RETIRE + 2 * B_PACKET_HALF * pk_per_step}
        pmc, src = pmc_traffic()
        kname = out["roofline"]["kernel"]
        if pmc and N == 65536 and kname in pmc and pmc[kname].get("launches", 0) >= 50:
            out["roofline"]["traffic"] = pmc[kname]["hbm_bytes_per_launch_raw"]
            out["roofline"]["traffic_source"] = src + " (raw FETCH_SIZE+WRITE_SIZE, KB units x 1024)"
            for other in out["roofline"].get("other_kernels", []):
                if other["kernel"] in pmc and pmc[other["kernel"]].get("launches", 0) >= 50:
                    other["traffic"] = pmc[other["kernel"]]["hbm_bytes_per_launch_raw"]
        if world == 1 and args.groups > 1:
            out["async_groups"] = async_groups(N, dev, K, W, args.groups)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
