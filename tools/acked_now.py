import os, sys, json
sys.path.insert(0, os.getcwd())
import numpy as np, torch, pcc_rl_amd
N=65536; dev=torch.device("cuda:0")
env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, record_steps=True)
gen = torch.Generator(device=dev).manual_seed(1234)
acts = torch.rand((400, N), generator=gen, device=dev) * 2 - 1
env.reset()
def leaves(n):
    if n <= 128: return 1 if n > 0 else 0
    if n > 8192: return leaves(8192) + leaves(n - 8192)
    n2 = n // 2; n2 -= n2 % 8
    return leaves(n2) + leaves(n - n2)
for t in range(301):
    obs, r, d, info = env.step(acts[t])
    if t in (20, 100, 200, 300):
        a = info["steps"][:, 1].cpu().numpy().astype(np.int64)
        q = np.quantile(a, [0.5, 0.75, 0.9, 0.97, 0.99, 0.999, 1.0]).tolist()
        top = np.sort(a)[-2048:]
        calls16 = [max(leaves(int(n)), leaves(int(n)//2) + leaves(int(n) - int(n)//2)) for n in top]
        print(json.dumps({"step": t, "acked_quantiles_50_75_90_97_99_999_max": q, "over_256": int((a > 256).sum()), "over_512": int((a > 512).sum()),
                          "over_1024": int((a > 1024).sum()), "over_4096": int((a > 4096).sum()),
                          "leaf_calls_16_lanes_top2048_p50_max": [int(np.median(calls16)), int(max(calls16))]}))
