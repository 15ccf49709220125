#!/usr/bin/env python3
"""Per-wavefront timeline of the send kernel at chosen steps of an episode (GPU box only).

    PCC_DEBUG_TIMELINE=1 python tools/send_timeline.py [n_envs] > gpurun_out/timeline.json

For each sampled step: kernel span, when the light wavefronts finished their lane rounds, how the
slowest wavefronts spent their time (rounds vs wave path), packets they carried."""
import json
import os
import sys

os.environ.setdefault("PCC_DEBUG_TIMELINE", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import pcc_rl_amd

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
S = int(sys.argv[2]) if len(sys.argv) > 2 else 1
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, n_senders=S)
for k, v in os.environ.items():
    if k.startswith("PCC_TUNE_"):
        env.set_tuning(**{k[9:].lower(): float(v)})
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N, S), generator=gen, device=dev) * 2 - 1   # one action vector per episode step, like bench.py
if os.environ.get("PCC_TL_ACTIONS") == "randn":   # what an untrained Gaussian policy sends (bench.py: policy_in_loop)
    acts = torch.randn((400, N, S), generator=gen, device=dev)
env.reset()
out = []
sample = {2, 10, 30, 60, 100, 150, 200, 250, 300, 350, 398}
for t in range(400):
    env.step_send(acts[t])
    if t in sample:
        raw = env.debug_timeline().astype(np.int64)
        tl = raw[:2 * N]             # item slots: light items from 0 (a range per partition), wave-path items from N (ditto), team items behind them
        fine = None
        if N >= 32768:
            # the light items' own progress (profile build): a stamp every 32 packets of the wavefront in slots 8192 + 2 s, 8193 + 2 s
            cand = np.nonzero((tl[:4096, 0] > 0) & (tl[:4096, 3] == 0))[0]
            if len(cand):
                t00 = min(tl[:8192][tl[:8192, 0] > 0][:, 0].min(), tl[N:][tl[N:, 0] > 0][:, 0].min() if (tl[N:, 0] > 0).any() else 1 << 62)
                last = cand[np.argsort(-tl[cand, 2])[:4]]
                fine = []
                for sl in last:
                    st_ = raw[8192 + 2 * sl: 8194 + 2 * sl].reshape(-1)[:16].astype(np.int64)
                    st_ = st_[(st_ >= tl[sl, 0]) & (st_ <= tl[sl, 2])]     # (stale stamps of earlier steps fall outside the item's span)
                    us = [round(float(v - t00) / 100.0, 2) for v in st_]
                    fine.append({"start_us": round(float(tl[sl, 0] - t00) / 100.0, 2), "finish_us": round(float(tl[sl, 2] - t00) / 100.0, 2),
                                 "largest_env": int(tl[sl, 5] & 0xFFFFFFFF), "stamp_us_every_32_packets": us,
                                 "ns_per_iteration_by_stretch": [round(1e3 * (us[k + 1] - us[k]) / 32.0) for k in range(len(us) - 1)]})
        if N >= 32768:
            tl = tl.copy()
            tl[8192:16384] = 0       # (the progress stamps above are not items)
        tl = tl[tl[:, 0] > 0]        # (the retire launch clears the slots: what is there belongs to this send launch)
        t0 = tl[:, 0].min()
        start, mid, fin = (tl[:, 0] - t0) / 100.0, (tl[:, 1] - t0) / 100.0, (tl[:, 2] - t0) / 100.0   # us
        heavy_wave = np.arange(len(tl))  # placeholder, the heavy wavefronts are the ones with 0 round time and w[3] > 0
        order = np.argsort(-fin)[:8]
        pct = lambda x: [float(np.percentile(x, p)) for p in (50, 90, 99, 100)]
        rec = {"step": t, "waves": int(len(tl)), "span_us": float(fin.max()),
               "start_us_p50_p90_p99_max": pct(start),
               "rounds_end_us": pct(mid), "finish_us": pct(fin),
               "packets_total": int(tl[:, 4].sum()), "wave_path_packets": int(tl[tl[:, 3] > 0, 6].sum()),
               # light items still running at 10-us marks of the launch (how many lane-round wavefronts share the machine in the tail)
               "light_items_running_at_us": {str(m): int(((tl[:, 3] == 0) & (start <= m) & (fin > m)).sum()) for m in range(10, 100, 10)},
               "wave_items_running_at_us": {str(m): int(((tl[:, 3] > 0) & (start <= m) & (fin > m)).sum()) for m in range(10, 100, 10)},
               "wave_path_envs": int(tl[:, 3].sum()),
               "busy_wave_us_total": float((fin - start).sum()),
               "slowest": [{"start": float(start[i]), "rounds_end": float(mid[i]), "finish": float(fin[i]),
                            "wave_path_envs": int(tl[i, 3]), "packets": int(tl[i, 4]), "largest_env": int(tl[i, 5] & 0xFFFFFFFF),
                            "wave_path_packets": int(tl[i, 6]) if tl[i, 3] > 0 else 0,
                            "first_round_us": float(tl[i, 6]) / 100.0 if tl[i, 3] == 0 else None,   # light items: start -> end of the first round (round_packets iterations)
                            "shader_clock_ghz": float((tl[i, 7] >> 32) / max(1e-9, (mid[i] - start[i]) * 1e3)) if tl[i, 3] == 0 else None,   # light items: cycles / wall time
                            "live": int(tl[i, 7] & 0xFF)} for i in order]}
        if os.environ.get("PCC_TL_RAW") and t in (100, 300):
            np.save(os.path.join(os.environ["PCC_TL_RAW"], "items_step%d.npy" % t), tl)
        rec["longest_light_items_progress"] = fine
        hv = tl[:, 3] > 0
        lt_ = ~hv
        dur = fin - start
        rec["heavy_items"] = {"n": int(hv.sum()), "busy_us": float(dur[hv].sum()), "packets": int(tl[hv, 4].sum()),
                              "ns_per_packet": float(1e3 * dur[hv].sum() / max(1, tl[hv, 4].sum())),
                              "ns_per_packet_p10_p50_p90": [float(np.percentile(1e3 * dur[hv] / np.maximum(1, tl[hv, 4]), q)) for q in (10, 50, 90)] if hv.any() else None}
        rec["light_items"] = {"n": int(lt_.sum()), "busy_us": float(dur[lt_].sum()), "packets": int(tl[lt_, 4].sum()),
                              "lane_iterations": int(tl[lt_, 5].sum()),
                              "ns_per_iteration": float(1e3 * dur[lt_].sum() / max(1, tl[lt_, 5].sum())),
                              "lane_efficiency": float(tl[lt_, 4].sum() / max(1, 64 * tl[lt_, 5].sum())),
                              "rounds_share": float((mid[lt_] - start[lt_]).sum() / max(1e-9, dur[lt_].sum()))}
        # the critical path of the launch: its span against the longest single item of each kind
        ih = int(np.argmax(np.where(hv, dur, -1.0))) if hv.any() else -1
        il = int(np.argmax(np.where(lt_, dur, -1.0))) if lt_.any() else -1
        rec["critical_path"] = {
            "launch_span_us": float(fin.max()),
            "longest_heavy_item": None if ih < 0 else {"us": float(dur[ih]), "start_us": float(start[ih]), "packets": int(tl[ih, 4]),
                                                       "ns_per_packet": float(1e3 * dur[ih] / max(1, tl[ih, 4]))},
            "heaviest_env": None if not hv.any() else {"packets": int(tl[hv, 4].max()),
                                                       "us": float(dur[hv][int(np.argmax(tl[hv, 4]))])},
            "longest_light_item": None if il < 0 else {"us": float(dur[il]), "start_us": float(start[il]),
                                                       "lane_iterations": int(tl[il, 5]), "ns_per_iteration": float(1e3 * dur[il] / max(1, tl[il, 5]))},
            "mean_busy_fraction_of_wavefronts": float(dur.sum() / (fin.max() * 256 * 16))}
        out.append(rec)
    env.step_retire()
print(json.dumps(out, indent=1))
