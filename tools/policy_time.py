#!/usr/bin/env python3
"""pcc_policy_act alone at 65 536 envs (GPU box): us per launch.  python tools/policy_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pcc_rl_amd.ppo import MlpPolicy
dev = torch.device("cuda:0")
torch.manual_seed(0)
pol = MlpPolicy(30, 1, (32, 16)).to(dev)
N = 65536
obs = torch.randn(N, 30, device=dev)
params = pol.flat_params()
noise = torch.randn(N, device=dev)
out = (torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev))
for _ in range(20):
    pol.act_fused(obs, True, params, noise, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(200):
    pol.act_fused(obs, True, params, noise, out)
e1.record()
torch.cuda.synchronize()
print("pcc_policy_act, 65 536 envs x 30 observations, 32-16 policy: %.2f us per launch" % (1e3 * e0.elapsed_time(e1) / 200))
a, logp, v = pol.act_fused(obs, False)
print("max |mean - framework| %.2e, max |value - framework| %.2e" % (float((a - pol.pi(obs)).abs().max()), float((v - pol.value(obs)).abs().max())))
