#!/usr/bin/env python3
"""Train a PCC rate controller with PPO on the GPU simulator.

    python examples/train_ppo.py --envs 8192 --iters 50 [--arch=32,16] [--gamma=0.99]

Counterpart of the reference's src/gym/stable_solve.py, but with the env, the rollout buffers and
the optimiser all on one MI355X (see pcc-rl_amd/ppo.py)."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pcc_rl_amd  # noqa: E402
from pcc_rl_amd.ppo import PPO  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=8192)
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--horizon", type=int, default=64)
    ap.add_argument("--arch", default="32,16")
    ap.add_argument("--gamma", type=float, default=0.99)
    ap.add_argument("--save", default="")
    args = ap.parse_args()
    env = pcc_rl_amd.BatchedNetworkEnv(args.envs, device="cuda:0", seed=0)
    agent = PPO(env, arch=tuple(int(x) for x in args.arch.split(",")), gamma=args.gamma, horizon=args.horizon)
    t0 = time.perf_counter()
    for it in range(args.iters):
        s = agent.iterate()
        steps = (it + 1) * args.envs * args.horizon
        print("iter %3d  env-steps %10d  reward/step %8.4f  entropy %6.3f  %.0f env-steps/s incl. learning"
              % (it, steps, s["mean_step_reward"], s["entropy"], steps / (time.perf_counter() - t0)))
    if args.save:
        torch.save(agent.policy.state_dict(), args.save)


if __name__ == "__main__":
    main()
