#!/usr/bin/env python3
"""What the whole training loop costs next to the env alone (GPU box only):
env-steps/s of (a) the env stepped with pre-generated actions, (b) the PPO rollout (policy forward +
sampling + env step + buffers), (c) rollout + GAE + the PPO epochs, all at the bench size."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pcc_rl_amd
from pcc_rl_amd.ppo import PPO

N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
T = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
# the reference's ratio: 8192 samples per iteration in minibatches of 2048 = 4 minibatches per epoch
agent = PPO(env, horizon=T, seed=0, minibatch=max(2048, N * T // 4))
acts = torch.rand((T, N), device=dev) * 2 - 1
def timed(fn, reps):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
def env_only():
    for t in range(T): env.step(acts[t])
t_env = timed(env_only, 3)
batch = {}
def rollout(): batch["b"] = agent.collect()
t_roll = timed(rollout, 3)
def update(): agent.update(*batch["b"][:5])
t_upd = timed(update, 2)
# the framework path of the same update (autograd + torch.optim.Adam): what the fused step replaces
agent_fw = PPO(env, horizon=T, seed=0, minibatch=max(2048, N * T // 4), fused_update=False)
def update_fw(): agent_fw.update(*batch["b"][:5])
t_upd_fw = timed(update_fw, 1)
# the same rollout double-buffered: the envs as two groups on their own streams, a group's policy kernel and env step queued
# on its stream, no join inside the rollout (PPO.collect over a GroupedNetworkEnv: bit-identical data, tests/test_ppo.py)
env.close()
del agent_fw
genv = pcc_rl_amd.GroupedNetworkEnv(N, 2, device=dev, seed=0)
if os.environ.get("PCC_GROUP_SEND_WAVES"):
    for e in genv.groups:
        e.set_tuning(send_waves=float(os.environ["PCC_GROUP_SEND_WAVES"]))
agent_g = PPO(genv, horizon=T, seed=0, minibatch=max(2048, N * T // 4))
gbatch = {}
def rollout_g(): gbatch["b"] = agent_g.collect()
t_roll_g = timed(rollout_g, 3)
def update_g(): agent_g.update(*gbatch["b"][:5])
t_upd_g = timed(update_g, 2)
out = {"n_envs": N, "horizon": T,
       "env_only": {"env_steps_per_s": N * T / t_env, "ms_per_step": 1e3 * t_env / T},
       "rollout": {"env_steps_per_s": N * T / t_roll, "ms_per_step": 1e3 * t_roll / T, "env_share_of_time": t_env / t_roll},
       "rollout_plus_update": {"env_steps_per_s": N * T / (t_roll + t_upd), "update_s": t_upd, "fused_update": agent.fused_update,
                               "env_share_of_time": t_env / (t_roll + t_upd)},
       "two_groups": {"rollout_env_steps_per_s": N * T / t_roll_g, "rollout_ms_per_step": 1e3 * t_roll_g / T, "update_s": t_upd_g,
                      "rollout_plus_update_env_steps_per_s": N * T / (t_roll_g + t_upd_g),
                      "note": "PPO.collect over GroupedNetworkEnv(N, 2): double-buffered sampling, the same data as one batch"},
       "framework_update": {"update_s": t_upd_fw, "rollout_plus_update_env_steps_per_s": N * T / (t_roll + t_upd_fw)},
       "note": "PPO with the reference script's policy shape and hyper-parameters (pi/vf MLP 32-16, minibatch 2048, 4 epochs); "
               "minibatch = N*T/4 keeps the reference's 4 minibatches per epoch (8192 samples / 2048) at this batch size"}
print(json.dumps(out, indent=1))
