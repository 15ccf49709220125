#!/usr/bin/env python3
"""Does WHERE a handle's allocations land decide how fast its launches are?  (GPU box.)  Several handles in a row in ONE process
and in fresh processes: the addresses of the ring tiers and the mean send / retire launch time over one episode each.
   python tools/placement.py [reps]"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def one(tag):
    import torch, pcc_rl_amd
    dev = torch.device("cuda:0")
    N = 65536
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
    gen = torch.Generator(device=dev).manual_seed(1234)
    acts = torch.rand((400, N, 1), generator=gen, device=dev) * 2 - 1
    env.reset()
    for t in range(20):
        env.step(acts[t])
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(380)]
    for k in range(380):
        ev[k][0].record(); env.step_send(acts[20 + k]); ev[k][1].record(); env.step_retire(); ev[k][2].record()
    torch.cuda.synchronize()
    send = sum(e[0].elapsed_time(e[1]) for e in ev[:-1]) / 379
    ret = sum(e[1].elapsed_time(e[2]) for e in ev[:-1]) / 379
    a = env.debug_addresses()
    env.close()
    return {"tag": tag, "send_ms": round(send, 4), "retire_ms": round(ret, 4), **{k: hex(v) for k, v in a.items()}}

if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        print(json.dumps([one("fresh process, handle %d" % k) for k in range(2)]))
        sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    out = []
    for r in range(reps):
        res = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        line = [l for l in res.stdout.splitlines() if l.startswith("[")]
        out += json.loads(line[-1]) if line else [{"error": res.returncode}]
    print(json.dumps(out, indent=1))
