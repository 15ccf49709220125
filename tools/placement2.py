#!/usr/bin/env python3
"""What puts a handle's retire launch into its fast or slow mode?  (GPU box.)  One fresh process per variant:
   plain      a handle, measured
   second     a handle created and closed unmeasured, then a handle measured
   prealloc   100 GB allocated through torch, written, freed (empty_cache), then a handle measured
   python tools/placement2.py [reps]"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def measure(env, torch, acts):
    env.reset()
    for t in range(20):
        env.step(acts[t])
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(380)]
    for k in range(380):
        ev[k][0].record(); env.step_send(acts[20 + k]); ev[k][1].record(); env.step_retire(); ev[k][2].record()
    torch.cuda.synchronize()
    return (round(sum(e[0].elapsed_time(e[1]) for e in ev[:-1]) / 379, 4), round(sum(e[1].elapsed_time(e[2]) for e in ev[:-1]) / 379, 4))

def child(variant):
    import torch, pcc_rl_amd
    dev = torch.device("cuda:0")
    N = 65536
    gen = torch.Generator(device=dev).manual_seed(1234)
    acts = torch.rand((400, N, 1), generator=gen, device=dev) * 2 - 1
    if variant == "prealloc":
        big = [torch.empty(25 * (1 << 30), dtype=torch.uint8, device=dev) for _ in range(4)]
        for b in big: b.fill_(1)
        torch.cuda.synchronize()
        del big
        torch.cuda.empty_cache()
    if variant == "second":
        e0 = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0); e0.reset(); e0.step(acts[0]); torch.cuda.synchronize(); e0.close()
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
    s, r = measure(env, torch, acts)
    env.close()
    print(json.dumps({"variant": variant, "send_ms": s, "retire_ms": r}))

if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "child":
        child(sys.argv[2]); sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    out = []
    for r in range(reps):
        for v in (sys.argv[2:] or ["plain", "second", "prealloc"]):
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "child", v], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
            line = [l for l in res.stdout.splitlines() if l.startswith("{")]
            out.append(json.loads(line[-1]) if line else {"variant": v, "error": res.returncode})
            print(out[-1], flush=True)
