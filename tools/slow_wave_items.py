#!/usr/bin/env python3
"""The slowest wave-path items of the send launch and the passes they ran (GPU box, profile build).
   PCC_DEBUG_TIMELINE=1 python tools/slow_wave_items.py [n_envs] [senders] [episodes] [rows]
A row per item: start and finish (us from the launch's first stamp), envs, packets, closed-form (token) passes, 256-packet sweep passes (two senders), chain passes,
plain-recurrence passes, ns per packet."""
import json, os, sys
os.environ.setdefault("PCC_DEBUG_TIMELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import pcc_rl_amd

WHY = {0: "tok:gap_tie", 1: "tok:t<260gap", 2: "tok:t_binade", 3: "tok:2tu<tend", 4: "tok:q=0", 5: "tok:e_range", 6: "tok:x0<=0", 7: "tok:eb>e",
       8: "tok:exp(tu)<e", 9: "tok:exp(maxq)<e", 10: "tok:grid", 16: "chain:tu<maxq", 17: "chain:gap_tie", 18: "chain:t<128gap", 19: "chain:t_binade",
       20: "chain:2tu<tend"}


def why_names(m):
    return [v for k, v in WHY.items() if (m >> k) & 1]


N = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
S = int(sys.argv[2]) if len(sys.argv) > 2 else 2
EPS = int(sys.argv[3]) if len(sys.argv) > 3 else 1     # episodes (auto-reset on): the link parameters are drawn again at every reset
TOP = int(sys.argv[4]) if len(sys.argv) > 4 else 12
STEPS = tuple(int(v) for v in os.environ.get("PCC_TL_STEPS", "100,200,300").split(","))   # steps of an episode to look at
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, n_senders=S)
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N, S), generator=gen, device=dev) * 2 - 1
env.reset()
if os.environ.get("PCC_TL_STAGGER"):   # envs out of lockstep like `bench.py --stagger`: env i starts its episode at pre-roll step i % 400
    phase = torch.arange(N, device=dev) % 400
    for t in range(400):
        if t:
            env.reset(phase == t)
        env.step(acts[t])
for tt in range(400 * EPS):
    t = tt % 400
    env.step_send(acts[t])
    if t in STEPS:
        raw = env.debug_timeline().astype(np.int64)
        tl = raw[:2 * N].copy()
        if N >= 32768:
            tl[8192:16384] = 0
        live = tl[:, 0] > 0
        t0 = tl[live, 0].min()
        wave = np.nonzero(live & (tl[:, 3] != 0))[0]
        light = np.nonzero(live & (tl[:, 3] == 0))[0]
        fin = (tl[:, 2] - t0) / 100.0
        rows = []
        for i in wave[np.argsort(-fin[wave])[:TOP]]:
            w7 = int(tl[i, 7])
            dur = (tl[i, 2] - tl[i, 0]) / 100.0
            rows.append({"start": round((tl[i, 0] - t0) / 100.0, 1), "finish": round(float(fin[i]), 1), "envs": int(tl[i, 3] & 0xFFFF),
                         "packets": int(tl[i, 4]), "closed": (w7 >> 8) & 0xFF, "sweep256": (w7 >> 16) & 0xFF, "chain": (w7 >> 24) & 0xFFFF, "plain": (w7 >> 40) & 0xFFFF,
                         "ns_per_packet": round(1e3 * dur / max(1, int(tl[i, 4])), 1), "refused_by": why_names(int(tl[i, 5]) >> 32)})
        # every single-env item: ns per packet against the share of its packets the chain sent (62 per pass)
        one = wave[(tl[wave, 3] & 0xFFFF) == 1]
        pk = tl[one, 4].astype(np.float64)
        dur = (tl[one, 2] - tl[one, 0]) / 100.0
        chain = ((tl[one, 7] >> 24) & 0xFFFF).astype(np.float64)
        closed = (((tl[one, 7] >> 8) & 0xFF) + ((tl[one, 7] >> 16) & 0xFF)).astype(np.float64)
        lrows = [{"start": round(float((tl[i, 0] - t0) / 100.0), 1), "rounds_end": round(float((tl[i, 1] - t0) / 100.0), 1), "finish": round(float(fin[i]), 1),
                  "packets": int(tl[i, 4]), "longest_env": int(tl[i, 5] & 0xFFFFFFFF)} for i in light[np.argsort(-fin[light])[:6]]] if len(light) else []
        print(json.dumps({"slowest_light_items": lrows, "light_items": int(len(light)),
                          "light_items_running_at_us": {str(u): int(((tl[light, 0] - t0) / 100.0 <= u).sum() - (fin[light] <= u).sum()) for u in (20, 40, 60, 80, 100, 120)}}), flush=True)
        print(json.dumps({"episode": tt // 400, "step": t, "span_us": round(float(fin[live].max()), 1), "light_last_us": round(float(fin[light].max()), 1) if len(light) else None,
                          "wave_items": int(len(wave)), "single_env_items": int(len(one)),
                          "single_env_us_total": round(float(dur.sum()), 0), "single_env_us_in_items_with_chain": round(float(dur[chain > 0].sum()), 0),
                          "single_env_items_with_chain": int((chain > 0).sum()),
                          "packets_per_closed_pass_p10_p50_p90": [round(float(v), 1) for v in np.percentile(pk[closed > 0] / closed[closed > 0], [10, 50, 90])] if (closed > 0).any() else None,
                          "refusals_over_items_with_chain": {v: int((((tl[one, 5] >> 32) >> k) & 1)[chain > 0].sum()) for k, v in WHY.items()},
                          "slowest": rows}), flush=True)
    env.step_retire()
