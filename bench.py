#!/usr/bin/env python3
"""Benchmark of the hot path: env steps/s of the batched simulator on MI355X.

    python bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[2], the one the metric is quoted on): 65 536 parallel envs
per GPU, single sender, per-env randomized bandwidth / latency / queue / loss over the ICML'19
ranges, actions U(-1, 1) pre-generated on the device, Philox loss uniforms, auto-reset at the
400-step episode boundary.  One "step" = one `step()` of all envs = one monitor interval per
env.  For N > 1 every rank owns its own 65 536 envs (weak scaling, env ids rank*65536+i, no
collective inside the step; episode returns are all-gathered over RCCL when episodes end).
`--gpus N` run directly (no torchrun) starts the N ranks itself.

Protocol: W warm-up steps, then K timed steps between barrier +
synchronize, repeated `--repeats` times back to back; the line reports the MEDIAN run, every run's
ms/step, and which steps of the 400-step episode each run covered (`window`).  The packet count per
step grows over an episode (about 120 early, 180 on average, 230 late), so WHERE a window shorter than
an episode sits decides what it measures: when K < 400 the script first runs one whole untimed-for-`value`
episode with per-step HIP events (reported as `whole_episode`: the section-8d metric), then places the K timed
steps -- after an untimed pre-roll -- on the stretch of the episode whose mean step time is closest to the
episode's mean, so that `value` estimates the whole-episode rate; `roofline` is computed from the whole episode.

`--config 2|3|5` picks the BASELINE.json configuration (default 3 = the one the metric is quoted on): 2 = 4 096 envs
on the fixed link (bw 200, 0.03 s, queue 5, no loss, rate0 60), 5 = 32 768 envs x 2 senders.

Prints ONE JSON line on rank 0 (see README / DESIGN.md section 6 for the fields).
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
# Algorithmic bytes per env-step: 450 B of fixed traffic + 32 B per packet
# (one 16-B in-flight record written at send, read at completion).  A step is two launches,
# timed apart with HIP events: send_kernel (dominant) reads the action (4), link parameters (32),
# and reads+writes link state (2x16) and sender rate/next_send/cursors (2x26) = 120 B, and writes
# the 16-B record; retire_kernel owns the remaining 330 B (state, history, obs/reward/done) and
# reads the record.
B_FIXED_SEND, B_FIXED_RETIRE, B_PACKET_HALF = 120, 330, 16
REFERENCE_STEPS_PER_S_PER_CORE = 800.0   # the unmodified reference, SURVEY.md section 6 (build container)


def spawn_ranks(args):
    """`python bench.py --gpus N` without torchrun: start the N ranks (one per GPU) ourselves."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(args.port or port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def code_stamp():
    """Which code this run measures: the commit the library was built from ("+" = csrc/ or include/ differed from it) and the hash
    of its sources (pcc-rl_amd/build.py leaves the stamp next to the library: the GPU box has the .so but no .git)."""
    try:
        from pcc_rl_amd import build as pbuild
        bi = pbuild.build_info(os.environ.get("PCC_SIM_LIBRARY"))
    except Exception:
        bi = {}
    if not bi:
        return "unknown"
    return "%s%s sources %s" % (bi.get("commit") or "no-git", "+" if bi.get("dirty") else "", bi.get("sources_sha16"))


def pmc_traffic():
    """HBM bytes per launch from the newest committed PMC summary (profiles/*_pmc_hbm.json): the fallback when the
    counter passes of this run (pmc_passes) are switched off or fail; labelled with its source and window."""
    pdir = os.path.join(ROOT, "profiles")
    names = sorted((n for n in os.listdir(pdir) if n.endswith("_pmc_hbm.json")), reverse=True) if os.path.isdir(pdir) else []
    for name in names:
        with open(os.path.join(pdir, name)) as f:
            return json.load(f), "profiles/" + name
    return None, None


def pmc_passes(config, timeout_s=150):
    """HBM traffic of this run's code on this run's GPU: two more runs of this script -- one whole episode each, no CPU
    baseline, nothing timed -- under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes with
    --kernel-trace only, as MI355X_MICROARCH.md prescribes), aggregated per kernel by tools/pmc_aggregate.py.  Returns
    (summary, None) or (None, why not)."""
    import shutil
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "this run is itself under a profiler"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import pmc_aggregate
    except ImportError as e:
        return None, "tools/pmc_aggregate.py: %s" % e
    tmp = tempfile.mkdtemp(prefix="pcc_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    env.setdefault("PCC_COMMIT", code_stamp())   # tools/pmc_aggregate.py stamps its summary with it
    dirs, line = [], None
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            d = os.path.join(tmp, counter.lower())
            cmd = [exe, "--pmc", counter, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
                   os.path.abspath(__file__), "--config", str(config), "--steps", "400", "--warmup", "20", "--repeats", "1",
                   "--no-cpu-baseline", "--no-pmc", "--no-policy", "--no-scaling"]
            try:
                r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=timeout_s)
            except subprocess.TimeoutExpired:
                return None, "the %s pass took more than %d s" % (counter, timeout_s)
            lines = [l for l in r.stdout.decode(errors="replace").splitlines() if l.startswith("{")]
            if r.returncode != 0 or not lines:
                return None, "the %s pass failed (exit code %d)" % (counter, r.returncode)
            line = lines[-1]
            dirs.append(d)
        return pmc_aggregate.aggregate(dirs, bench_line=line), None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(seconds_budget=15.0):
    """The CPU side of the same workload on this host's cores, a bounded sample each:
    the C oracle (oracle/pcc_oracle.c, the literal heap-based restatement of the reference engine)
    on one thread and on every core, and the same algorithm in the reference's own language
    (oracle/pcc_oracle_py.py: heapq + numpy), one process per core."""
    import numpy as np

    import oracle

    cores = os.cpu_count() or 1
    rs = np.random.RandomState(0)
    n_envs, n_steps = 64, 100
    done_steps, done_pk, t_used, base = 0, 0.0, 0.0, 0
    while t_used < seconds_budget:
        acts = rs.uniform(-1, 1, (n_envs, n_steps))
        t0 = time.perf_counter()
        out = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=0, env_gid_base=base, n_threads=1, want_obs=False)
        t_used += time.perf_counter() - t0
        done_steps += n_envs * n_steps
        done_pk += float(out["steps"][..., 0].sum())
        base += n_envs
    res = {"value": done_steps / t_used, "unit": "env steps/s", "cores": 1, "kind": "port",
           "sample": "%d env-steps (%d-env x %d-step batches of the bench workload, Philox uniforms, "
                     "%.1f packets/step) on the C oracle, 1 thread, %.1f s; host has %d cores"
                     % (done_steps, n_envs, n_steps, done_pk / done_steps, t_used, cores)}
    if cores > 1:   # the same C oracle on every host core (envs are independent: one env batch per thread)
        n_all, steps_all, t_all, base_all = 8 * cores, 0, 0.0, 1 << 20
        while t_all < 5.0:
            acts = rs.uniform(-1, 1, (n_all, n_steps))
            t0 = time.perf_counter()
            oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=0, env_gid_base=base_all, n_threads=cores, want_obs=False)
            t_all += time.perf_counter() - t0
            steps_all += n_all * n_steps
            base_all += n_all
        res["all_cores"] = {"value": steps_all / t_all, "unit": "env steps/s", "cores": cores,
                            "sample": "%d env-steps on %d threads, %.1f s" % (steps_all, cores, t_all)}
    # the reference's algorithm class in its own language (heapq + numpy): one process on every second PHYSICAL core, pinned
    # -- the figure to hold against the reference's ~800 steps/s/core (within ~30 % on comparable
    # silicon).  One process per logical CPU (round 3: 256 of them) measures SMT siblings and a saturated memory system
    # instead: 300 per core; one process alone on an idle host measures the turbo clock of one core: 4 500.
    code = ("import sys, json, os; sys.path.insert(0, %r)\n"
            "cpu = int(sys.argv[2])\n"
            "if cpu >= 0:\n"
            "    try: os.sched_setaffinity(0, {cpu})\n"
            "    except OSError: pass\n"
            "from oracle.pcc_oracle_py import time_episodes\n"
            "tot_s = tot_p = tot_t = 0.0; k = 0\n"
            "while tot_t < 10.0:\n"
            "    s, p, t = time_episodes(int(sys.argv[1]) * 1000 + k, 1, n_steps=200); tot_s += s; tot_p += p; tot_t += t; k += 1\n"
            "print(json.dumps([tot_s, tot_p, tot_t]))\n" % ROOT)
    model, phys = "unknown", {}
    try:
        with open("/proc/cpuinfo") as f:
            cur = {}
            for line in f:
                if ":" in line:
                    k, v = [x.strip() for x in line.split(":", 1)]
                    cur[k] = v
                elif cur:
                    phys.setdefault((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))), int(cur.get("processor", 0)))
                    model = cur.get("model name", model)
                    cur = {}
            if cur:
                phys.setdefault((cur.get("physical id", "0"), cur.get("core id", cur.get("processor"))), int(cur.get("processor", 0)))
                model = cur.get("model name", model)
    except OSError:
        pass
    allowed = set(os.sched_getaffinity(0))
    first_cpus = sorted(c for c in phys.values() if c in allowed) or sorted(allowed)
    pinned = first_cpus[::2] or first_cpus          # every second physical core
    res["host"] = {"cpu_model": model, "logical_cpus": cores, "physical_cores": len(phys) or None}
    procs = [subprocess.Popen([sys.executable, "-c", code, str(r), str(c)], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL)
             for r, c in enumerate(pinned)]
    t0 = time.perf_counter()
    outs = []
    for p in procs:
        try:
            outs.append(json.loads(p.communicate()[0].decode() or "[0,0,1]"))
        except ValueError:
            outs.append([0, 0, 1])
    wall = time.perf_counter() - t0
    per_core = [o[0] / o[2] for o in outs if o[0] > 0]
    # ... and ONE process with the machine to itself
    try:
        alone = json.loads(subprocess.run([sys.executable, "-c", code, "7777", "-1"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                          timeout=60).stdout.decode() or "[0,0,1]")
    except (ValueError, subprocess.TimeoutExpired):
        alone = [0, 0, 1]
    if per_core:
        agg = sum(o[0] for o in outs) / wall
        mean_pc = sum(per_core) / len(per_core)
        ratio = mean_pc / REFERENCE_STEPS_PER_S_PER_CORE
        res["python_port"] = {"value": agg, "unit": "env steps/s", "cores": len(per_core),
                              "per_core": mean_pc, "vs_reference_per_core": ratio,
                              "within_30_percent_of_the_reference": bool(0.7 <= ratio <= 1.3),
                              "one_process_alone": alone[0] / alone[2],
                              "one_process_alone_vs_reference": (alone[0] / alone[2]) / REFERENCE_STEPS_PER_S_PER_CORE,
                              "sample": "one process on every second physical core, pinned (%d processes on a %s, %d physical cores) "
                                        "x >= 10 s of 200-step default-parameter episodes (oracle/pcc_oracle_py.py: heapq + numpy, "
                                        "%.1f packets/step), %.1f s wall; the unmodified reference measured %.0f steps/s/core in the "
                                        "build container (SURVEY.md section 6: an 8-core host of unknown model, under the survey's "
                                        "own load) -- a ratio outside 0.7..1.3 says how this host's cores compare with those, not "
                                        "that the port does different work (tests/test_oracle_py.py: bit-equal results)"
                                        % (len(per_core), model, len(phys), sum(o[1] for o in outs) / max(1.0, sum(o[0] for o in outs)),
                                           wall, REFERENCE_STEPS_PER_S_PER_CORE)}
    return res


def policy_in_loop(pcc_rl_amd, torch, N, dev, horizon=64):
    """The real use of the env, timed in the same run: env-steps/s of PPO.collect() -- the fused policy kernel (pcc_policy_act: both
    MLPs, the Gaussian sample from N(0, 1) noise, log-probability, value) writing the action row the env's step reads, `horizon`
    steps -- and of rollout + GAE + the PPO epochs (pcc_ppo_minibatch_step), at the bench size.  An untrained policy's N(0, 1)
    actions spread the rates further than the bench's U(-1, 1) do, so the env's own launches are slower here than in `value`."""
    from pcc_rl_amd.ppo import PPO
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
    agent = PPO(env, horizon=horizon, seed=0, minibatch=max(2048, N * horizon // 4))
    box = {}

    def timed(fn, reps):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps

    recorded = []

    def rollout():
        box["b"] = agent.collect()
        recorded.append(box["b"][1])     # the actions the policy chose: [horizon, N, 1]

    t_roll = timed(rollout, 3)
    t_upd = timed(lambda: agent.update(*box["b"][:5]), 2)
    env.check_flags()
    # ---- where a rollout step's time goes (round 6): (a) the env alone under the SAME actions from the same state -- a second
    # handle of the same seed replays the four recorded rollouts (the first untimed, like above), its two launches timed apart on
    # every 7th step; (b) the policy kernel alone on the last rollout's observation rows; (c) the rest = launch gaps / the
    # rollout's own bookkeeping (noise, GAE, the value of the last observation)
    split = None
    try:
        obs_rows = box["b"][0]
        env.close()
        env = None
        env2 = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0)
        env2.reset()
        for t in range(horizon):
            env2.step(recorded[0][t])
        n_rep = (len(recorded) - 1) * horizon
        ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(3)] for k in range(0, n_rep, 7)}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_rep):
            a = recorded[1 + k // horizon][k % horizon]
            e = ev.get(k)
            if e: e[0].record()
            env2.step_send(a)
            if e: e[1].record()
            env2.step_retire()
            if e: e[2].record()
        torch.cuda.synchronize()
        t_env = (time.perf_counter() - t0) / n_rep
        env2.check_flags()
        env2.close()
        params = agent.policy.flat_params()
        noise = torch.randn((horizon, N), device=dev)
        a_o, l_o, v_o = torch.empty(N, device=dev), torch.empty(N, device=dev), torch.empty(N, device=dev)
        for t in range(4):
            agent.policy.act_fused(obs_rows[t], True, params, noise[t], (a_o, l_o, v_o))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for t in range(horizon):
            agent.policy.act_fused(obs_rows[t], True, params, noise[t], (a_o, l_o, v_o))
        torch.cuda.synchronize()
        t_pol = (time.perf_counter() - t0) / horizon
        split = {"env_alone_under_the_policys_actions_ms": 1e3 * t_env,
                 "env_send_ms": sum(e[0].elapsed_time(e[1]) for e in ev.values()) / len(ev),
                 "env_retire_ms": sum(e[1].elapsed_time(e[2]) for e in ev.values()) / len(ev),
                 "policy_kernel_ms": 1e3 * t_pol,
                 "rest_ms": 1e3 * (t_roll / horizon - t_env - t_pol),
                 "note": "a second handle (same seed) replays the recorded actions of the timed rollouts from the same state: what the env's "
                         "two launches cost under N(0,1)-sampled actions (they spread the sending rates further than the line's U(-1,1) "
                         "does: more packets in the largest envs); policy_kernel_ms = pcc_policy_act back to back on the rollout's "
                         "observation rows; rest = launch gaps, the rollout's noise / GAE / last-value launches"}
    except Exception as e:
        split = {"error": "%s: %s" % (type(e).__name__, e)}
    if env is not None:
        env.close()
    return {"rollout": {"value": N * horizon / t_roll, "unit": "env steps/s", "ms_per_step": 1e3 * t_roll / horizon, "split": split},
            "rollout_plus_update": {"value": N * horizon / (t_roll + t_upd), "unit": "env steps/s", "update_s": t_upd,
                                    "fused_update": bool(agent.fused_update)},
            "envs": N, "horizon": horizon,
            "note": "PPO.collect() over %d envs x %d steps (fused policy kernel + step_into, N(0,1)-sampled actions of the untrained "
                    "32-16 policy), then GAE + 4 epochs x 4 minibatches of the fused gradient step; 3 / 2 repetitions after one "
                    "untimed; supplementary -- `value` is the env with pre-generated U(-1,1) actions (SURVEY.md section 8d)" % (N, horizon)}


def scaling_in_n(pcc_rl_amd, torch, dev, sizes=(131072, 262144), max_steps=400):
    """Supplementary: the same workload (config 3's links and action law) at LARGER batches per GPU -- the step's two launches each
    wait for their own longest work item, so a larger batch fills the wavefront slots the headline size leaves idle.  One whole
    episode per size after 40 warm-up steps, HIP events around the two launches on every 7th step."""
    out = []
    for n in sizes:
        env = None
        try:
            env = pcc_rl_amd.BatchedNetworkEnv(n, device=dev, seed=0, auto_reset=True, max_steps=max_steps)
            gen = torch.Generator(device=dev).manual_seed(1234)
            acts = torch.rand((max_steps, n, 1), generator=gen, device=dev, dtype=torch.float32) * 2 - 1
            env.reset()
            for t in range(40):
                env.step(acts[t])
            ev = {k: [torch.cuda.Event(enable_timing=True) for _ in range(3)] for k in range(0, max_steps, 7)}
            sent0 = env.state("total_sent").sum()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(max_steps):
                e = ev.get(k)
                if e: e[0].record()
                env.step_send(acts[(40 + k) % max_steps])
                if e: e[1].record()
                env.step_retire()
                if e: e[2].record()
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            env.check_flags()
            ks = [k for k in ev if (40 + k + 1) % max_steps != 0]
            out.append({"envs": n, "value": n * max_steps / el, "unit": "env steps/s", "ms_per_step": 1e3 * el / max_steps,
                        "send_ms": sum(ev[k][0].elapsed_time(ev[k][1]) for k in ks) / len(ks),
                        "retire_ms": sum(ev[k][1].elapsed_time(ev[k][2]) for k in ks) / len(ks),
                        "packets_per_env_step": float((env.state("total_sent").sum() - sent0).item()) / (n * max_steps),
                        "device_bytes": int(env.device_bytes) if hasattr(env, "device_bytes") else None})
        except Exception as e:   # (supplementary: never in the way of the line)
            out.append({"envs": n, "error": "%s: %s" % (type(e).__name__, e)})
        finally:
            if env is not None:
                env.close()
            torch.cuda.empty_cache()
    return out


def async_groups(pcc_rl_amd, torch, N, dev, K, W, n_groups=4):
    """Supplementary figure, NOT the headline: the same N envs as independent groups on their own
    streams, stepped without a per-step synchronization between the groups (double-buffered
    sampling: the policy works on one group while the others simulate)."""
    K = min(K, 400)
    env = pcc_rl_amd.GroupedNetworkEnv(N, n_groups, device=dev, seed=0)
    acts = []
    for g in range(n_groups):
        gen = torch.Generator(device=dev).manual_seed(4321 + g)
        acts.append(torch.rand((64, env.group_size), generator=gen, device=dev, dtype=torch.float32) * 2 - 1)
    env.reset()
    for t in range(W):
        for g in range(n_groups):
            env.step_group(g, acts[g][t % 64])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(W, W + K):
        for g in range(n_groups):
            env.step_group(g, acts[g][t % 64])
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    env.check_flags()
    env.close()
    return {"value": N * K / el, "unit": "env steps/s", "groups": n_groups, "steps": K,
            "note": "same %d envs as %d independent groups on their own streams, no per-step sync between groups; "
                    "supplementary, the headline value is one synchronous batch" % (N, n_groups)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed steps per run (5 episodes by default)")
    ap.add_argument("--warmup", type=int, default=50)
    ap.add_argument("--repeats", type=int, default=3, help="timed runs, back to back; the line reports the median")
    ap.add_argument("--envs", type=int, default=0, help="envs per GPU (default: the configuration's size)")
    ap.add_argument("--config", type=int, default=3, choices=(2, 3, 5),
                    help="BASELINE.json configuration: 3 (default) = 65 536 envs, randomized links, 1 sender; 2 = 4 096 envs on the "
                         "fixed link; 5 = 32 768 envs x 2 senders on one bottleneck")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--port", type=int, default=0, help="rendezvous port when bench.py starts the ranks itself")
    ap.add_argument("--ring-capacity", type=int, default=0, help="records per accepted ring (0 = library default)")
    ap.add_argument("--no-policy", action="store_true",
                    help="skip the supplementary policy_in_loop figures (PPO rollout and rollout + update at the bench size)")
    ap.add_argument("--no-scaling", action="store_true",
                    help="skip the supplementary scaling_in_n figures (one episode each at 131 072 and 262 144 envs per GPU)")
    ap.add_argument("--no-pmc", action="store_true",
                    help="skip the two rocprofv3 counter passes that measure roofline.traffic (HBM bytes per launch)")
    ap.add_argument("--groups", type=int, default=0,
                    help="also measure the same envs as this many independent groups on their own streams "
                         "(supplementary field async_groups)")
    ap.add_argument("--event-stride", type=int, default=7,
                    help="HIP events around the two kernels on every this-many-th timed step (recording an event costs the "
                         "stream ~4.5 us, three per step are 7 %% of a step of config 3: the timed region carries them on a "
                         "sample of its steps -- 7 is coprime to the episode length, so every episode phase is sampled over a run -- "
                         "1 = every step)")
    ap.add_argument("--fused", action="store_true",
                    help="step a full-size batch by the one-launch step (step_fused_kernel, PCC_TUNE_FUSED: an env's retire half follows "
                         "its own send half inside the launch; measured slower, off by default) instead of a send launch and a retire "
                         "launch (pcc_step_send / pcc_step_retire, timed apart)")
    ap.add_argument("--stagger", action="store_true",
                    help="spread the envs' episode phases uniformly over the 400 steps before timing (masked resets "
                         "during an untimed pre-roll): every window then sees the episode-average load, and the "
                         "auto-resets run on the device-gated path instead of the host-known lockstep boundary")
    ap.add_argument("--max-steps", type=int, default=400, help="episode length (tests shorten it to see the all-gather)")
    ap.add_argument("--share-device", action="store_true",
                    help="testing only: every rank uses cuda:0 (lets the N > 1 path run on a 1-GPU box with gloo)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ:
        raise SystemExit(spawn_ranks(args))

    import torch

    import pcc_rl_amd
    from pcc_rl_amd import distributed as pdist

    rank, world, local_rank = pdist.rank_info()
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s) (WORLD_SIZE)" % (args.gpus, world))
    if args.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    numa = pdist.pin_to_gpu_numa_node(local_rank) if world > 1 and not args.share_device else None
    dist = None
    if world > 1:
        import torch.distributed as dist
        pdist.init_process_group(args.backend, device=dev)   # "nccl" is RCCL on ROCm

    cfg = args.config
    N = args.envs or {2: 4096, 3: 65536, 5: 32768}[cfg]
    S = 2 if cfg == 5 else 1
    K, W, R = args.steps, args.warmup, max(1, args.repeats)
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=dev, seed=0, env_gid_base=pdist.env_gid_base(rank, N), n_senders=S,
                                       link_params=(200.0, 0.03, 5.0, 0.0, 60.0) if cfg == 2 else None,   # ns:459-464
                                       auto_reset=True, ring_capacity=args.ring_capacity, max_steps=args.max_steps,
                                       # experiments: "d1,d2,d3" = pool divisors for pcc_set_ring_pools (tools/r06/pmc_attrib.sh)
                                       ring_pools=[int(v) for v in os.environ["PCC_BENCH_RING_POOLS"].split(",")] if os.environ.get("PCC_BENCH_RING_POOLS") else None)
    if os.environ.get("PCC_BENCH_TUNING"):   # experiments: {"knob": value, ...} for env.set_tuning (speed only, never results)
        env.set_tuning(**json.loads(os.environ["PCC_BENCH_TUNING"]))
    gen = torch.Generator(device=dev).manual_seed(1234 + rank)
    # one action vector per step of an episode (fresh U(-1, 1) actions every step), generated before the timed
    # region.  (A short pool is not harmless: cycled, it gives every env the same net rate change each cycle, the rates
    # drift to their limits and the cost of a step depends on the phase of the cycle -- 0.116 to 0.161 ms per send
    # launch for a pool of 64, profiles/r03_experiments.json.)
    pool = args.max_steps
    actions = torch.rand((pool, N, S), generator=gen, device=dev, dtype=torch.float32) * 2 - 1
    env.reset()
    max_steps = env.max_steps
    if args.stagger:
        # env i starts its episode at pre-roll step i % 400: after 400 steps the phases are uniform
        phase = torch.arange(N, device=dev) % max_steps
        for s in range(max_steps):
            if s:
                env.reset(phase == s)
            env.step(actions[s % pool])
    returns_gathered = 0
    gather_buf = torch.empty((world * N,), dtype=torch.float32, device=dev) if world > 1 else None

    small = N < 8192   # the library steps a batch this small in ONE launch (step_small_kernel): no halves to time apart
    if args.fused:
        env.set_tuning(fused=1)
    # one launch per step: a small batch, or (--fused) the fused step of a full-size one (step_fused_kernel; out of lockstep --
    # --stagger -- the library steps by two launches inside the same call)
    fused = small or (args.fused and not args.stagger)

    def one_step(t, ev=None):
        nonlocal returns_gathered
        if ev is not None:
            ev[0].record()
        if fused:
            env.step(actions[t % pool])
        else:
            env.step_send(actions[t % pool])
        if ev is not None:
            ev[1].record()
        if not fused:
            env.step_retire()
        if ev is not None:
            ev[2].record()
        if world > 1 and (t + 1) % max_steps == 0:
            # the only inter-GPU traffic on this path: episode returns, once per episode
            pdist.gather_episode_returns(env.episode_returns().to(torch.float32), out=gather_buf)
            returns_gathered += 1

    t_global = 0
    whole = None
    if K < max_steps and not args.stagger:
        # ---- whole episodes first (the section-8d metric), timed in segments of 5 steps: where in the episode do K steps
        # take the episode-mean time per step?  (All ranks do the same work; rank 0's times place the window.)
        # (episodes differ -- a handful of envs whose queue limit sits just above a power of two send on the slow exact
        # path and can be a launch's critical path: 0.113 / 0.150 / 0.125 ms per send launch for the first three episodes of
        # seed 0 -- so the figure is taken over kWhole episodes)
        kWhole = 3
        sent0 = env.state("total_sent").sum()
        # Events are not free (an event record costs the stream ~4.5 us): a boundary event every SEG steps places the window,
        # and every HALVES-th step carries the two more that time its send and its retire launch apart
        SEG, HALVES = (5 if max_steps % 5 == 0 else 1), 14
        n_all = kWhole * max_steps
        ev_b = [torch.cuda.Event(enable_timing=True) for _ in range(n_all // SEG + 1)]
        ev_h = {k: [torch.cuda.Event(enable_timing=True) for _ in range(3)] for k in range(0, n_all, HALVES)}
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        c0 = time.perf_counter()
        for k in range(n_all):
            if k % SEG == 0:
                ev_b[k // SEG].record()
            one_step(t_global, ev_h.get(k))
            t_global += 1
        ev_b[n_all // SEG].record()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = pdist.max_over_ranks(time.perf_counter() - c0, device=dev)   # MAX over ranks, like the timed runs below
        seg_t = [ev_b[j].elapsed_time(ev_b[j + 1]) / SEG for j in range(n_all // SEG)]       # ms per step, by segment
        per_ep = max_steps // SEG
        # mean over the episodes, by step of the episode (a step takes its segment's mean)
        step_t = [sum(seg_t[e * per_ep + k // SEG] for e in range(kWhole)) / kWhole for k in range(max_steps)]
        mean_t = sum(step_t[:-SEG]) / (max_steps - SEG)        # (the last segment also runs the episode-boundary reset)
        halves = sorted(ev_h)
        if fused:   # (the step after a reset runs without work lists, as two launches; the last one of an episode also runs the reset)
            halves = [k for k in halves if k % max_steps != 0 and (k + 1) % max_steps != 0]
        send_s = [ev_h[k][0].elapsed_time(ev_h[k][1]) for k in halves]
        ret_s = [ev_h[k][1].elapsed_time(ev_h[k][2]) for k in halves if (k + 1) % max_steps != 0]
        span = W + K
        best, best_err = 0, None
        csum = [0.0]
        for v in step_t:
            csum.append(csum[-1] + v)
        for st in range(0, max_steps - span):
            err = abs((csum[st + span] - csum[st + W]) / K - mean_t)
            if best_err is None or err < best_err:
                best, best_err = st, err
        whole = {"ms_per_step": 1e3 * el / (kWhole * max_steps), "value": world * N * kWhole * max_steps / el, "episodes": kWhole,
                 "send_ms": sum(send_s) / len(send_s), "retire_ms": sum(ret_s) / len(ret_s), "steps_with_kernel_events": len(send_s),
                 "packets_per_env_step": float((env.state("total_sent").sum() - sent0).item()) / (N * kWhole * max_steps),
                 "window_start": best}
        if world > 1:   # every rank must pre-roll alike
            b = torch.tensor([best], device=dev)
            dist.broadcast(b, 0)
            best = int(b.item())
        for _ in range(best):                                   # untimed pre-roll to the chosen stretch of the next episode
            one_step(t_global)
            t_global += 1
    for _ in range(W):
        one_step(t_global)
        t_global += 1

    runs = []
    # HIP events on the launch stream (torch's current stream IS the stream the library launches on); all of them are
    # made before the first run and the packet counters stay on the device until the last one is over, so that the GPU
    # does not sit idle (and clock down) between the warm-up steps and a timed region that may be only a few ms long
    ES = max(1, args.event_stride)
    all_ev = [[[torch.cuda.Event(enable_timing=True) for _ in range(3)] if k % ES == 0 else None for k in range(K)] for _ in range(R)]
    sent_marks = [env.state("total_sent").sum()]
    for r in range(R):
        ev = all_ev[r]
        first = t_global
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(K):
            one_step(t_global, ev[k] if k % ES == 0 else None)
            t_global += 1
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        sent_marks.append(env.state("total_sent").sum())
        runs.append({"elapsed": elapsed, "first_step": first})
    for r in range(R):
        ev, first = all_ev[r], runs[r]["first_step"]
        runs[r]["per_rank_ms_per_step"] = [1e3 * v / K for v in pdist.gather_floats(runs[r]["elapsed"], device=dev)]
        runs[r]["elapsed"] = pdist.max_over_ranks(runs[r]["elapsed"], device=dev)     # MAX over ranks (bench contract)
        runs[r]["packets"] = float((sent_marks[r + 1] - sent_marks[r]).item())
        # steps that also ran the episode-boundary reset kernels are kept out of the retire average
        timed_k = [k for k in range(K) if k % ES == 0]   # the steps that carry events
        plain = [k for k in timed_k if (first + k + 1) % max_steps != 0] if not args.stagger else timed_k
        if fused and not args.stagger:
            timed_k = [k for k in plain if (first + k) % max_steps != 0] or timed_k
        runs[r]["send_ms"] = sum(ev[k][0].elapsed_time(ev[k][1]) for k in timed_k) / len(timed_k)
        runs[r]["retire_ms"] = sum(ev[k][1].elapsed_time(ev[k][2]) for k in plain) / max(1, len(plain))
        runs[r]["event_steps"] = len(timed_k)
        runs[r]["first_steps_ms"] = [ev[k][0].elapsed_time(ev[k][2]) for k in timed_k[:6]]
    many = None
    if small and world == 1 and not args.stagger:
        # a small batch: one step is a 25 us launch, less than a trip around the Python loop above.  Supplementary: the same K
        # steps queued by ONE library call (pcc_step_many: the loop runs in C); `value` stays the one-call-per-step figure
        acts_many = torch.stack([actions[(t_global + k) % pool] for k in range(K)])
        env.step_many(acts_many[:8])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        env.step_many(acts_many)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        many = {"value": N * K / el, "unit": "env steps/s", "us_per_step": 1e6 * el / K, "steps": K,
                "note": "the same steps queued from C by one pcc_step_many call (open loop: actions pre-computed); supplementary"}
    if not os.environ.get("PCC_BENCH_IGNORE_FLAGS"):   # experiments only: an overflowed ring means invalid results
        env.check_flags()
    restart_stats = env.restart_stats() if args.stagger else None
    fused_steps = env.fused_steps()
    env.close()
    coll = pdist.collective_info()
    numa_all = None
    if world > 1:
        numa_all = [None] * world
        dist.all_gather_object(numa_all, numa)

    if rank == 0:
        order = sorted(range(R), key=lambda r: runs[r]["elapsed"])
        med = runs[order[R // 2]]
        value = world * N * K / med["elapsed"]
        ms_per_step = 1e3 * med["elapsed"] / K
        window_fields = None
        # `value` / `ms_per_step` / `timed_steps` / `repeats` / `runs_ms_per_step` describe ONE measurement: value = envs x timed_steps /
        # (ms_per_step x timed_steps).  `steps` stays the K the caller asked for (the bench contract echoes its arguments).
        timed_steps, timed_repeats, timed_runs = K, R, [1e3 * r["elapsed"] / K for r in runs]
        if whole is not None:
            # fewer timed steps than an episode: the K-step window has no episode boundary inside, and SURVEY.md section 8d's
            # metric includes the auto-resets -- so `value` is the whole episodes' figure (three 400-step episodes with their
            # boundary resets, timed by this run just before the window), and the window is reported next to it
            window_fields = {"value": value, "ms_per_step": ms_per_step, "steps": K, "repeats": R,
                             "runs_ms_per_step": [1e3 * r["elapsed"] / K for r in runs],
                             "note": "the %d timed steps after the warm-up (the bench contract's K), placed where an episode has its mean "
                                     "step time; no episode boundary inside" % K}
            value, ms_per_step = whole["value"], whole["ms_per_step"]
            timed_steps, timed_repeats, timed_runs = whole["episodes"] * max_steps, 1, [whole["ms_per_step"]]
        pk_per_step = med["packets"] / (N * K)
        send_ms, retire_ms = med["send_ms"], med["retire_ms"]
        roof_src = "the timed steps"
        if whole is not None:   # the roofline of a short run is the whole episode's (the timed window is K steps of it)
            pk_roof, send_ms, retire_ms = whole["packets_per_env_step"], whole["send_ms"], whole["retire_ms"]
            roof_src = "the whole episodes run before the timed steps (whole_episode)"
        else:
            pk_roof = pk_per_step
        # (two senders: the fixed bytes of section 8d once per sender)
        send_bytes = N * (B_FIXED_SEND * S + B_PACKET_HALF * pk_roof)
        retire_bytes = N * (B_FIXED_RETIRE * S + B_PACKET_HALF * pk_roof)
        if args.stagger:
            window = {"episode_steps": "all phases at once (--stagger): every step sees the episode-average load"}
        else:
            f = med["first_step"] % max_steps
            window = {"episode_steps": [f, f + K - 1] if K < max_steps else "whole episodes",
                      "first_step_of_episode": f, "steps": K,
                      "covers_whole_episodes": K % max_steps == 0,
                      "note": None if K >= max_steps else
                      "steps < one 400-step episode: packets per step grow over an episode (about 120 early, 180 on average, "
                      "230 late); the window was placed (untimed pre-roll) where a whole episode run just before had its mean "
                      "step time within %d steps -- `whole_episode` is that episode's own figure" % (W + K)}
        out = {
            "metric": "env steps/sec (whole node) at 64k parallel envs",
            "value": value, "unit": "env steps/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": ms_per_step, "timed_steps": timed_steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "repeats": timed_repeats, "runs_ms_per_step": timed_runs,
            "median_run_kernel_ms": {"send": med["send_ms"], "retire": med["retire_ms"], "first_steps": med["first_steps_ms"],
                                     "steps_with_events": med["event_steps"], "event_stride": ES,
                                     "note": "HIP events around the two kernels on every event_stride-th timed step (an event record "
                                             "costs the stream ~4.5 us; three on every step are 7 % of a step of config 3)"},
            "spread": (max(r["elapsed"] for r in runs) - min(r["elapsed"] for r in runs)) / med["elapsed"],
            "window": window,
            "code": code_stamp(),
            "config": {"workload": ("BASELINE config %d: %d envs/GPU, %s, U(-1,1) actions, 400-step episodes, auto-reset%s"
                                    % (cfg, N, {2: "1 sender, fixed link (bw 200 pkt/s, 0.03 s, queue 5, no loss, rate0 60)",
                                                3: "1 sender, per-env randomized bw/latency/queue/loss (ICML'19 ranges)",
                                                5: "2 senders on one bottleneck, per-env randomized links (ICML'19 ranges)"}[cfg],
                                       ", episode phases staggered" if args.stagger else "")),
                       "baseline_config": cfg, "envs_per_gpu": N, "senders": S, "packets_per_env_step": pk_per_step,
                       "episode_return_allgathers": returns_gathered},
        }
        # what proves the multi-GPU line: the process group's own world size and backend (n_gpus above is the launcher's), the
        # RCCL version, every rank's own time for the timed steps (value uses their MAX), where each rank was pinned
        out["distributed"] = {"backend": coll["backend"], "dist_world_size": coll["dist_world_size"], "rccl_version": coll["rccl_version"],
                              "per_rank_ms_per_step": med.get("per_rank_ms_per_step"), "rank_cpu_binding": numa_all,
                              "launcher": "torchrun" if os.environ.get("TORCHELASTIC_RUN_ID") else "plain"}
        if window_fields is not None:
            out["timed_window"] = window_fields
            out["value_source"] = ("whole_episode: %d whole %d-step episodes with their boundary resets (timed_steps = %d, one pass, MAX over "
                                   "ranks), run and timed before the %d-step window (--steps < one episode); the K-step window with its own "
                                   "repeats is `timed_window`" % (whole["episodes"], max_steps, whole["episodes"] * max_steps, K))
        if whole is not None:
            out["whole_episode"] = dict(whole, unit="env steps/s", steps=whole["episodes"] * max_steps,
                                        note="%d whole %d-step episodes run before the timed steps (a boundary event every 5 steps, the "
                                             "kernels' own events on every 14th): the section-8d metric and, with --steps below one "
                                             "episode, the line's `value`; the K-step window (`timed_window`) sits at episode step "
                                             "window_start + warmup of the next episode" % (whole["episodes"], max_steps))
        send_gbps = send_bytes / (send_ms * 1e-3) / 1e9
        retire_gbps = retire_bytes / (retire_ms * 1e-3) / 1e9
        both = (send_bytes + retire_bytes) / ((send_ms + retire_ms) * 1e-3) / 1e9
        if fused:   # one kernel: its bytes are both halves', its time is what the first event pair bracketed
            send_bytes, retire_bytes = send_bytes + retire_bytes, 0.0
            send_gbps = send_bytes / (send_ms * 1e-3) / 1e9
            retire_gbps, retire_ms = 0.0, 0.0
            both = send_gbps
        out["roofline"] = {"bound": "hbm", "kernel": ("step_small_kernel<%d, false>" if small else "step_fused_kernel<%d, false>" if fused else "send_kernel<%d, false>") % S, "achieved": send_gbps,
                           "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": send_gbps / HBM_PEAK_GBPS,
                           "traffic": None, "kernel_ms": send_ms, "algorithmic_bytes_per_launch": send_bytes,
                           "measured_over": roof_src,
                           "other_kernels": [] if fused else [{"kernel": "retire_kernel<%d, false>" % S, "achieved": retire_gbps,
                                              "frac": retire_gbps / HBM_PEAK_GBPS, "kernel_ms": retire_ms,
                                              "algorithmic_bytes_per_launch": retire_bytes}],
                           "whole_step": {"achieved": both, "frac": both / HBM_PEAK_GBPS}}
        if not fused and not small and S == 1:
            # what the dominant kernel's traffic IS: scattered 16-byte ring appends, one per packet.  The chip absorbs ~205 G of
            # them per second (3.3 TB/s) whatever the coding -- measured by tools/microbench/lane_round (variants 2 / 7 / 8:
            # one record per request, four per request through LDS, four per request by a drain wavefront) and round 4's
            # store_bench4 -- and the send launch's busy phase (its first ~55 us, before the serial tail) runs at 196 G/s.
            out["roofline"]["scattered_record_ceiling"] = {
                "records_per_s": 2.05e11, "GB/s": 3280.0, "frac_of_hbm_peak": 3280.0 / HBM_PEAK_GBPS,
                "launch_average_records_per_s": N * pk_roof / (send_ms * 1e-3),
                "launch_average_frac_of_ceiling": N * pk_roof / (send_ms * 1e-3) / 2.05e11,
                "note": "supplementary: the rate at which the memory system takes scattered 16-byte records = PARTIAL-LINE writes "
                        "(profiles/r06_store_bw.txt: 3.0-3.25 TB/s; whole lines, dense or scattered, 5.1-6.6 TB/s; "
                        "profiles/r06_lane_round_microbench.txt); only the lane rounds' records (45 % of a launch's) are written that "
                        "way since round 6, the wave path's leave as whole lines; the launch average includes the serial tail of the "
                        "longest lane-round items (profiles/r06_send_tail.json), during which that bandwidth idles"}
        # HBM traffic from the counters: measured now, by two profiled runs of this script (unless --no-pmc / not one GPU);
        # else the newest committed summary, labelled as such
        kname = out["roofline"]["kernel"]
        pmc, src, why = None, None, "--no-pmc"
        if not args.no_pmc and world == 1:
            pmc, why = pmc_passes(cfg)
            src = "measured by this run"
        live = pmc is not None
        if pmc is None and N == 65536 and cfg == 3:
            pmc, src = pmc_traffic()
        if pmc and kname in pmc and pmc[kname].get("launches", 0) >= 50 and "hbm_bytes_per_launch" in pmc[kname]:
            out["roofline"]["traffic"] = pmc[kname]["hbm_bytes_per_launch"]
            out["roofline"]["traffic_over_algorithmic"] = pmc[kname].get("traffic_over_algorithmic")
            what = ("2 x FETCH_SIZE + WRITE_SIZE (KB units x 1024; the factor 2 is this repo's calibration, "
                    "profiles/r02_pmc_calibration.json), rocprofv3 --pmc in separate passes")
            if live:
                out["roofline"]["traffic_source"] = (src + ": " + what + " over two more runs of this script on this GPU, one whole "
                                                     "episode each (after 20 warm-up steps); traffic_over_algorithmic compares it with "
                                                     "THAT window's algorithmic bytes")
            else:
                out["roofline"]["traffic_source"] = (src + " (the counter passes of this run: " + str(why) + "): " + what + " of a SEPARATE "
                                                     "profiled run of this script (code at commit " + str(pmc.get("_commit", "?")) + ") over " +
                                                     str(pmc.get("_window", "steps 20..120 of an episode")) + "; compare it with that "
                                                     "window's algorithmic bytes (in the profile), not this run's")
            for other in out["roofline"].get("other_kernels", []):
                if other["kernel"] in pmc and pmc[other["kernel"]].get("launches", 0) >= 50 and "hbm_bytes_per_launch" in pmc[other["kernel"]]:
                    other["traffic"] = pmc[other["kernel"]]["hbm_bytes_per_launch"]
                    other["traffic_over_algorithmic"] = pmc[other["kernel"]].get("traffic_over_algorithmic")
        elif why:
            out["roofline"]["traffic_source"] = "none: " + str(why)
        if restart_stats is not None:
            out["restarts_out_of_lockstep"] = restart_stats
        out["config"]["step_launches"] = ("one (step_small_kernel)" if small else "one (step_fused_kernel: an env's retire half follows its own send "
                                          "half inside the launch); %d of this handle's steps ran that way, the others -- the step after "
                                          "each reset, steps out of lockstep -- as send + retire launches" % fused_steps if fused else
                                          "two (send_kernel, retire_kernel)")
        if many is not None:
            out["many_steps_per_call"] = many
        if world == 1 and cfg == 3 and not args.stagger and not args.no_policy and N >= 8192:
            try:
                out["policy_in_loop"] = policy_in_loop(pcc_rl_amd, torch, N, dev)
            except Exception as e:   # (supplementary: never in the way of the line)
                out["policy_in_loop"] = {"error": "%s: %s" % (type(e).__name__, e)}
        if world == 1 and cfg == 3 and not args.stagger and not args.no_scaling and N == 65536 and not fused:
            pts = scaling_in_n(pcc_rl_amd, torch, dev, max_steps=max_steps)
            head = {"envs": N, "value": value, "unit": "env steps/s", "ms_per_step": ms_per_step, "send_ms": send_ms, "retire_ms": retire_ms,
                    "packets_per_env_step": pk_roof, "note": "the line's own figures"}
            out["scaling_in_n"] = {"points": [head] + pts,
                                   "note": "envs per GPU beyond BASELINE's 65 536 (same links, same action law, one whole episode each after 40 "
                                           "warm-up steps; supplementary): the two launches of a step are as long as their longest work "
                                           "items, so a larger batch costs less than proportionally -- what a trainer should run per GPU"}
        if world == 1 and args.groups > 1:
            out["async_groups"] = async_groups(pcc_rl_amd, torch, N, dev, K, W, args.groups)
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        elif world == 1:
            out["cpu_baseline"] = None
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
