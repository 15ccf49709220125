#!/usr/bin/env python3
"""A learning curve that means something (GPU box): train the reference's PPO set-up on the GPU simulator for a few hundred
million env-steps, then hold the trained rate controller against baselines on envs it has never seen.

    python examples/learning_curve.py [--envs 8192] [--env-steps 2e8] [--eval-envs 4096] > profiles/r05_learning_curve.json

Training: `pcc_rl_amd.PPO` (policy shape and hyper-parameters of the reference's src/gym/stable_solve.py:30-58), whole-episode
iterations (horizon = 400 steps: every iteration averages the same mix of episode phases).  The reference trains for
6 x 1600 x 410 = 3.9e6 env-steps (stable_solve.py:54-58); the default here is ~50 times that.

Evaluation: `--eval-envs` held-out envs (another seed and another range of env ids: other links, other loss draws), one whole
400-step episode each, the SAME envs for every controller:
  trained        the trained policy's mean action (no exploration noise)
  untrained      the same network before training (mean action)
  fixed_rate     action 0 at every step: the sender stays at its starting rate, U(0.3, 1.5) x bandwidth (ns:466)
  aimd           additive increase of 1 % of the starting rate per interval, halving the rate after an interval that lost packets
                 (the textbook TCP-like rule, computed from the env's own per-step record: a baseline, not part of the library)
Reported per controller: mean episode return (the quantity PPO maximises, ns:194,205), loss ratio (lost / sent over the episode),
latency ratio (mean over steps of avg latency / connection minimum, so:183-188) and link utilisation (acknowledged packets per
second of simulated time / bandwidth)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcc_rl_amd  # noqa: E402
from pcc_rl_amd.ppo import PPO  # noqa: E402

COL_SENT, COL_ACKED, COL_LOST, COL_RATE, COL_TIME, COL_RUN_DUR, COL_REWARD, COL_M0 = 0, 1, 2, 3, 4, 5, 6, 7
M_LATENCY_RATIO = 10


def evaluate(name, act_fn, n, dev, seed, gid_base):
    """One whole episode of n held-out envs under act_fn(obs, env, t) -> actions [n]."""
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=dev, seed=seed, env_gid_base=gid_base, record_steps=True, auto_reset=False)
    obs = env.reset()
    bw = env.state("bw")
    t0 = env.state("now").clone()
    ret = torch.zeros(n, dtype=torch.float64, device=dev)
    sent = torch.zeros(n, dtype=torch.float64, device=dev)
    acked, lost, lat_ratio = torch.zeros_like(sent), torch.zeros_like(sent), torch.zeros_like(sent)
    last = None
    for t in range(env.max_steps):
        a = act_fn(obs, env, t, last)
        obs, r, d, info = env.step(a)
        rows = info["steps"]
        last = rows
        ret += rows[:, COL_REWARD]
        sent += rows[:, COL_SENT]
        acked += rows[:, COL_ACKED]
        lost += rows[:, COL_LOST]
        lat_ratio += rows[:, COL_M0 + M_LATENCY_RATIO]
    dur = env.state("now") - t0
    env.check_flags()
    out = {"controller": name, "envs": n, "mean_episode_return": float(ret.mean()), "median_episode_return": float(ret.median()),
           "loss_ratio": float(lost.sum() / sent.sum()), "mean_latency_ratio": float((lat_ratio / env.max_steps).mean()),
           "link_utilisation": float((acked / dur / bw).mean()), "packets_per_env_step": float(sent.sum() / (n * env.max_steps))}
    env.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--env-steps", type=float, default=2e8)
    ap.add_argument("--eval-envs", type=int, default=4096)
    ap.add_argument("--horizon", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    env = pcc_rl_amd.BatchedNetworkEnv(args.envs, device=dev, seed=args.seed)
    agent = PPO(env, horizon=args.horizon, seed=args.seed)
    untrained = {k: v.clone() for k, v in agent.policy.state_dict().items()}
    iters = max(1, int(round(args.env_steps / (args.envs * args.horizon))))
    curve = []
    t0 = time.perf_counter()
    for it in range(iters):
        s = agent.iterate()
        curve.append({"iteration": it, "env_steps": (it + 1) * args.envs * args.horizon,
                      "mean_episode_return": s["mean_step_reward"] * env.max_steps, "mean_step_reward": s["mean_step_reward"],
                      "entropy": s.get("entropy"), "wall_s": time.perf_counter() - t0})
    train_s = time.perf_counter() - t0
    env.check_flags()
    env.close()

    pol = agent.policy
    held = dict(n=args.eval_envs, dev=dev, seed=args.seed + 12345, gid_base=1 << 24)

    def policy_mean(obs, env_, t, last):
        with torch.no_grad():
            return pol.pi(obs).reshape(-1)

    results = [evaluate("trained", policy_mean, **held)]
    trained_state = {k: v.clone() for k, v in pol.state_dict().items()}
    pol.load_state_dict(untrained)
    results.append(evaluate("untrained", policy_mean, **held))
    pol.load_state_dict(trained_state)
    results.append(evaluate("fixed_rate", lambda obs, e, t, last: torch.zeros(e.n_envs, device=dev), **held))

    def aimd(obs, e, t, last):
        # rate <- rate + 1 % of the starting rate, or rate / 2 after an interval with a loss; as the env's action:
        # rate * (1 + 0.025 a) going up, rate / (1 - 0.025 a) going down (ns:235-241)
        rate = e.state("rate")[0]
        rate0 = e.state("rate0")[0]
        up = (0.01 * rate0 / rate) / 0.025
        down = torch.full_like(up, -40.0)               # rate / (1 + 1)
        if last is None:
            return up.to(torch.float32)
        return torch.where(last[:, COL_LOST] > 0, down, up).to(torch.float32)

    results.append(evaluate("aimd", aimd, **held))
    by = {r["controller"]: r for r in results}
    out = {"what": "PPO (reference hyper-parameters, pi/vf MLP 32-16) on the MI355X simulator, then the trained controller against "
                   "baselines on held-out envs; python examples/learning_curve.py",
           "training": {"envs": args.envs, "horizon": args.horizon, "iterations": iters, "env_steps": iters * args.envs * args.horizon,
                        "reference_budget_env_steps": 6 * 1600 * 410, "wall_s": train_s,
                        "env_steps_per_s_incl_learning": iters * args.envs * args.horizon / train_s,
                        "first_iteration_return": curve[0]["mean_episode_return"], "last_iteration_return": curve[-1]["mean_episode_return"],
                        "best_iteration_return": max(c["mean_episode_return"] for c in curve)},
           "held_out_evaluation": results,
           "trained_vs": {k: {"return_gain": by["trained"]["mean_episode_return"] - by[k]["mean_episode_return"]}
                          for k in ("untrained", "fixed_rate", "aimd")},
           "curve": curve}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
