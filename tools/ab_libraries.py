#!/usr/bin/env python3
"""Two builds of the library against each other (GPU box): send and retire launch times at 65 536 envs (AB_ENVS, AB_SENDERS), steps 20..400 of an
episode, HIP events around each launch, one fresh process per run, interleaved (a handle's retire launch has a fast and a
slow mode of its own: profiles/r05_placement.json).
   python tools/ab_libraries.py [reps] path/libA.so path/libB.so ..."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import torch, pcc_rl_amd
    dev = torch.device("cuda:0")
    N = int(os.environ.get("AB_ENVS", "65536"))
    NS = int(os.environ.get("AB_SENDERS", "1"))
    gen = torch.Generator(device=dev).manual_seed(1234)
    acts = torch.rand((400, N, NS) if NS > 1 else (400, N, 1), generator=gen, device=dev) * 2 - 1
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, n_senders=NS)
    env.reset()
    for t in range(20):
        env.step(acts[t])
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(380)]
    for k in range(380):
        ev[k][0].record(); env.step_send(acts[20 + k]); ev[k][1].record(); env.step_retire(); ev[k][2].record()
    torch.cuda.synchronize()
    s = sum(e[0].elapsed_time(e[1]) for e in ev[:-1]) / 379
    r = sum(e[1].elapsed_time(e[2]) for e in ev[:-1]) / 379
    env.check_flags()
    env.close()
    print(json.dumps({"library": os.path.basename(os.environ.get("PCC_SIM_LIBRARY", "libpcc_sim.so")), "send_ms": round(s, 4),
                      "retire_ms": round(r, 4), "step_ms": round(s + r, 4)}))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child(); sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    libs = sys.argv[2:]
    out = []
    for r in range(reps):
        for lib in libs:
            e = dict(os.environ, PCC_SIM_LIBRARY=os.path.abspath(lib))
            res = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=e)
            line = [l for l in res.stdout.splitlines() if l.startswith("{")]
            out.append(json.loads(line[-1]) if line else {"library": lib, "error": res.returncode, "stderr": res.stderr[-400:]})
            print(json.dumps(out[-1]), flush=True)
    by = {}
    for o in out:
        if "step_ms" in o:
            by.setdefault(o["library"], []).append(o)
    print(json.dumps({k: {m: round(sum(o[m] for o in v) / len(v), 4) for m in ("send_ms", "retire_ms", "step_ms")} for k, v in by.items()}))
