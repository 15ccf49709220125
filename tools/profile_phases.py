#!/usr/bin/env python3
"""Per-wavefront phase breakdown of the step kernel (diagnostics; GPU box only).

Uses pcc_set_profile_buffer: every lane stores shader-clock stamps at the phase boundaries of
its monitor interval.  Lanes of a wave run in lockstep, so a wave's stamps are its phase times.
Writes a JSON summary (per-phase cycle totals, critical-path wave, packets-per-lane stats)."""
import ctypes
import json
import sys
import os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcc_rl_amd  # noqa: E402
from pcc_rl_amd import native  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    warm = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    out_path = sys.argv[3] if len(sys.argv) > 3 else "gpurun_out/phases.json"
    dev = torch.device("cuda:0")
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
    env.reset()
    gen = torch.Generator(device=dev).manual_seed(1)
    for _ in range(warm):
        env.step(torch.rand((N,), generator=gen, device=dev) * 2 - 1)
    prof = torch.zeros((N, 8), dtype=torch.int64, device=dev)
    native.check(env._L.pcc_set_profile_buffer(env._h, ctypes.c_void_p(prof.data_ptr())))
    res = []
    for it in range(5):
        a = torch.rand((N,), generator=gen, device=dev) * 2 - 1
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        env.step(a)
        e1.record()
        torch.cuda.synchronize()
        p = prof.cpu().double()
        w = p.view(N // 64, 64, 8)
        ts = w[:, 0, :6]                       # lockstep: lane 0 speaks for the wave
        phases = {
            "pass0_streams": (ts[:, 1] - ts[:, 0]),
            "send_stream": (ts[:, 2] - ts[:, 1]),
            "pass1_streams": (ts[:, 3] - ts[:, 2]),
            "mi_end_event": (ts[:, 4] - ts[:, 3]),
            "metrics_obs": (ts[:, 5] - ts[:, 4]),
        }
        total = ts[:, 5] - ts[:, 0]
        sent = w[:, :, 6]
        retired = w[:, :, 7]
        span = float(ts[:, 5].max() - ts[:, 0].min())
        res.append({
            "kernel_ms": e0.elapsed_time(e1),
            "span_cycles_first_start_to_last_end": span,
            "wave_cycles_mean": float(total.mean()), "wave_cycles_max": float(total.max()),
            "wave_cycles_p50": float(total.median()),
            "phase_share_of_sum": {k: float(v.sum() / total.sum()) for k, v in phases.items()},
            "critical_wave_phases": {k: float(v[total.argmax()]) for k, v in phases.items()},
            "sent_per_lane_mean": float(sent.mean()), "sent_per_lane_max": float(sent.max()),
            "sent_wave_max_mean": float(sent.max(dim=1).values.mean()),
            "retired_wave_max_mean": float(retired.max(dim=1).values.mean()),
            "cycles_per_wave_iteration_send": float(phases["send_stream"].sum() / sent.max(dim=1).values.sum()),
        })
    with open(out_path, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res[-1], indent=1))


if __name__ == "__main__":
    main()
