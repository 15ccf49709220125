#!/usr/bin/env python3
"""Shape of the fused step's launch at chosen steps of an episode (GPU box only; profile build):

    PCC_DEBUG_TIMELINE=1 python tools/fused_timeline.py [n_envs] > gpurun_out/fused_timeline.json

Per sampled step: when the send items start and end (light items / wave-path items), when the retire units are claimed,
ready and done, how long a claimed unit waited for its envs, busy wavefront time of either half, and -- the point of the
one-launch step -- how much of the retire work was done before the last send item ended."""
import json
import os
import sys

os.environ.setdefault("PCC_DEBUG_TIMELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pcc_rl_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
for k, v in os.environ.items():
    if k.startswith("PCC_TUNE_"):
        env.set_tuning(**{k[9:].lower(): float(v)})
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N, 1), generator=gen, device=dev) * 2 - 1
env.reset()
out = []
sample = {int(x) for x in os.environ.get("PCC_TL_STEPS", "5,30,100,200,300,398").split(",")}
pct = lambda x: [float(np.percentile(x, p)) for p in (10, 50, 90, 99, 100)] if len(x) else None
for t in range(max(sample) + 1):
    env.step(acts[t])
    if t not in sample:
        continue
    raw = env.debug_timeline().astype(np.int64)
    items = raw[:2 * N]
    rblocks = (N + 7) // 8 + 1
    head = raw[2 * N + 2 * rblocks]                          # (the first row of the region: running counter, launch start, sequence number)
    units = raw[2 * N + 2 * rblocks + 4:].reshape(-1, 4)    # (rows 2, 3 of the region: the XCDs of the two launches' first blocks)
    tag = units[:, 0] >> 32
    cur = units[tag == int(head[2])]
    q = (cur[:, 0] >> 31) & 1
    envs = cur[:, 0] & 0x7FFFFFFF    # envs of the unit
    claim, ready, done = cur[:, 1], cur[:, 2], cur[:, 3]
    # this launch's send items: the slots are not cleared between launches; block 0 stamps the launch's start (the others start within a microsecond)
    t0 = int(head[1]) - 100
    it = items[(items[:, 0] >= t0) & (items[:, 2] >= items[:, 0])]
    us = lambda x: (x - t0) / 100.0
    light = it[it[:, 3] == 0] if len(it) else it
    wavep = it[it[:, 3] > 0] if len(it) else it
    send_end = float(us(it[:, 2]).max())
    unit_us = (done - ready) / 100.0
    wait_us = (ready - claim) / 100.0
    done_us = us(done)
    order = np.argsort(done_us)
    rec = {"step": t, "launch_span_us": float(done_us.max()),
           "send": {"items": int(len(it)), "light_items": int(len(light)), "wave_path_items": int(len(wavep)),
                    "start_us_p10_p50_p90_p99_max": pct(us(it[:, 0])), "end_us_p10_p50_p90_p99_max": pct(us(it[:, 2])),
                    "light_end_us": pct(us(light[:, 2])) if len(light) else None,
                    "wave_path_end_us": pct(us(wavep[:, 2])) if len(wavep) else None,
                    "last_send_item_ends_us": send_end,
                    "busy_wave_us": float(((it[:, 2] - it[:, 0]) / 100.0).sum())},
           "retire": {"units": int(len(cur)), "units_16_lanes": int((q == 1).sum()), "envs": int(envs.sum()),
                      "claimed_us_p10_p50_p90_p99_max": pct(us(claim)), "ready_us": pct(us(ready)), "done_us": pct(done_us),
                      "unit_us_8_lanes": pct(unit_us[q == 0]), "unit_us_16_lanes": pct(unit_us[q == 1]),
                      "wait_for_envs_us": pct(wait_us), "waited_total_us": float(wait_us.sum()),
                      "busy_wave_us": float(unit_us.sum()),
                      "first_unit_starts_us": float(us(ready).min()),
                      "units_done_before_the_last_send_item_ends": int((done_us <= send_end).sum()),
                      "share_of_retire_time_before_the_last_send_item_ends":
                          float(np.clip(np.minimum(done_us, send_end) - us(ready), 0, None).sum() / max(1e-9, unit_us.sum()))},
           "wave_slots": 4096,
           "busy_fraction": float((((it[:, 2] - it[:, 0]) / 100.0).sum() + unit_us.sum()) / (done_us.max() * 4096))}
    # retire throughput over time: units done per 10 us
    hist, _ = np.histogram(done_us, bins=np.arange(0, done_us.max() + 10, 10))
    rec["retire"]["units_done_per_10us"] = [int(v) for v in hist]
    hist2, _ = np.histogram(us(it[:, 2]), bins=np.arange(0, done_us.max() + 10, 10))
    rec["send"]["items_done_per_10us"] = [int(v) for v in hist2]
    out.append(rec)
env.check_flags()
print(json.dumps({"fused_steps": env.fused_steps(), "steps": out}, indent=1))
