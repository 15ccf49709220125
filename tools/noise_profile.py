#!/usr/bin/env python3
"""USE_LATENCY_NOISE at 16 384 envs, a few steps: run under `rocprofv3 --kernel-trace --stats` to see which kernel the time is in
(the two instances of noise_sorted_kernel, retire_kernel<1, true>).  (GPU box.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, pcc_rl_amd
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
K = int(sys.argv[2]) if len(sys.argv) > 2 else 40
dev = torch.device("cuda:0")
S = int(sys.argv[3]) if len(sys.argv) > 3 else 1
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, latency_noise=1.1, n_senders=S)
if os.environ.get("NOISE_SORTED"):
    env.set_tuning(noise_sorted=int(os.environ["NOISE_SORTED"]))
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((K, N, S), generator=gen, device=dev) * 2 - 1
env.reset()
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for t in range(K):
    env.step(acts[t])
torch.cuda.synchronize()
print("ms per step", (time.perf_counter() - t0) / K * 1e3)
env.check_flags()
