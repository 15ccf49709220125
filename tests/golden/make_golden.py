#!/usr/bin/env python3
"""Generate golden input/output vectors from the UNMODIFIED reference env.

Runs only in the build container (needs /root/reference); the GPU box never
sees the reference, only the .npz files this script writes next to itself.

Recipe:
  * a ~25-line stand-in for the four `gym` symbols network_sim.py touches is put
    into sys.modules (gym is not installed here),
  * /root/reference/src/gym goes on sys.path and `network_sim` is imported as-is,
  * per env, `network_sim.random = random.Random(seed)` routes every draw of the
    module-global RNG (5 per constructor, 5 per reset, 1 per sent packet) to a
    private MT19937 stream,
  * actions come from `np.random.RandomState(seed)`,
  * after every step the sender counters, clock, run_dur, reward and all 12
    monitor-interval metrics are recorded.

Nothing from the reference is copied: the fixtures hold inputs (seed, link
parameters, actions) and outputs (per-step numbers) only.

Usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
"""
import io
import os
import random
import sys
import tempfile
import types
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src/gym"

METRICS = ["send rate", "recv rate", "recv dur", "send dur", "avg latency",
           "loss ratio", "ack latency inflation", "sent latency inflation",
           "conn min latency", "latency increase", "latency ratio", "send ratio"]
DEFAULT_FEATURES = "sent latency inflation,latency ratio,send ratio"


def install_gym_standin():
    gym = types.ModuleType("gym")

    class Env(object):
        pass

    class Box(object):
        def __init__(self, low, high, dtype=np.float32):
            self.low, self.high, self.dtype = np.asarray(low), np.asarray(high), dtype
            self.shape = self.low.shape

    spaces = types.ModuleType("gym.spaces")
    spaces.Box = Box
    utils = types.ModuleType("gym.utils")
    seeding = types.ModuleType("gym.utils.seeding")
    seeding.np_random = lambda seed=None: (np.random.RandomState(seed), seed)
    utils.seeding = seeding
    envs = types.ModuleType("gym.envs")
    registration = types.ModuleType("gym.envs.registration")
    registration.register = lambda **kw: None
    envs.registration = registration
    gym.Env, gym.spaces, gym.utils, gym.envs = Env, spaces, utils, envs
    for name, mod in [("gym", gym), ("gym.spaces", spaces), ("gym.utils", utils),
                      ("gym.utils.seeding", seeding), ("gym.envs", envs),
                      ("gym.envs.registration", registration)]:
        sys.modules[name] = mod


def import_reference():
    sys.dont_write_bytecode = True
    install_gym_standin()
    sys.path.insert(0, REF)
    with contextlib.redirect_stdout(io.StringIO()):
        import network_sim  # noqa: E402  (the unmodified reference module)
    return network_sim


class CountingRandom(random.Random):
    """MT19937 stream that counts draws, so fixtures can say where in the stream
    of `random.Random(seed)` an episode's per-packet draws begin."""

    def __init__(self, seed):
        super().__init__(seed)
        self.n = 0

    def random(self):
        self.n += 1
        return super().random()


class Recorder(object):
    """Collects one episode of per-step records from a reference env."""

    def __init__(self, n_senders=1):
        self.rows = []
        self.obs = []

    def row(self, ns, env_or_none, net, sender, reward, run_dur, obs):
        mi = sender.get_run_data()
        vals = [mi.get(m) for m in METRICS]
        self.rows.append([sender.sent, sender.acked, sender.lost, sender.rate,
                          net.cur_time, run_dur, reward] + [float(v) for v in vals])
        self.obs.append(np.asarray(obs, dtype=np.float64).copy())


def run_env_episodes(ns, seed, n_steps, action_fn, history_len=10,
                     features=DEFAULT_FEATURES, fixed=None, n_episodes=1, cwnd=False, noise=False):
    """One reference env object driven for n_episodes; returns dict of arrays.

    cwnd=True runs the engine with its dormant USE_CWND option on (ns:54; a module global the
    engine reads at call time, set here from outside -- the reference file is not modified):
    actions are then [rate action, cwnd action] pairs (ns:376-377, 413-414).
    noise=True does the same with USE_LATENCY_NOISE (ns:51; MAX_LATENCY_NOISE stays the reference's 1.1)."""
    rng = CountingRandom(seed)
    ns.random = rng
    ns.USE_CWND = bool(cwnd)
    ns.USE_LATENCY_NOISE = bool(noise)
    env = ns.SimulatedNetworkEnv(history_len=history_len, features=features)
    offsets = []
    orig_create = env.create_new_links_and_senders

    def counted_create():
        orig_create()
        offsets.append(rng.n)
    env.create_new_links_and_senders = counted_create
    if fixed is not None:
        bw, lat, queue, loss, rate0 = fixed

        def fixed_links():
            env.links = [ns.Link(bw, lat, queue, loss), ns.Link(bw, lat, queue, loss)]
            env.senders = [ns.Sender(rate0, [env.links[0], env.links[1]], 0,
                                     env.features, history_len=env.history_len)]
            env.run_dur = 3 * lat
            offsets.append(rng.n)
        env.create_new_links_and_senders = fixed_links
    out = []
    rs = np.random.RandomState(seed)
    for ep in range(n_episodes):
        obs0 = env.reset()
        link = env.links[0]
        sender = env.senders[0]
        params = [link.bw, link.dl, link.max_queue_delay * link.bw, link.lr,
                  sender.starting_rate, env.run_dur]
        warm = [env.net.cur_time, len(env.net.q)]
        actions = action_fn(rs, n_steps)
        rec = Recorder()
        dones = []
        cwnds = []
        for t in range(n_steps):
            obs, reward, done, info = env.step(actions[t] if cwnd else [actions[t]])
            rec.row(ns, env, env.net, sender, reward, env.run_dur, obs)
            dones.append(done)
            cwnds.append(sender.cwnd)
        out.append(dict(params=np.array(params, dtype=np.float64),
                        queue=int(round(params[2])),
                        warm=np.array(warm, dtype=np.float64),
                        rng=np.array([offsets[-1], rng.n], dtype=np.int64),
                        obs0=np.asarray(obs0, dtype=np.float64),
                        actions=np.asarray(actions, dtype=np.float64),
                        steps=np.array(rec.rows, dtype=np.float64),
                        cwnd=np.array(cwnds, dtype=np.int64),
                        obs=np.array(rec.obs, dtype=np.float64),
                        done=np.array(dones, dtype=np.bool_)))
    ns.USE_CWND = False
    ns.USE_LATENCY_NOISE = False
    return out


def pack(cases, keep_full_obs=2):
    """Stack a list of episode dicts into one npz-able dict."""
    d = {}
    d["seed"] = np.array([c["seed"] for c in cases], dtype=np.int64)
    d["episode"] = np.array([c["episode"] for c in cases], dtype=np.int64)
    d["params"] = np.stack([c["params"] for c in cases])       # bw, dl, queue, lr, rate0, run_dur0
    d["warm"] = np.stack([c["warm"] for c in cases])           # cur_time, heap length after warm-up
    d["rng"] = np.stack([c["rng"] for c in cases])             # draws before first packet, draws at episode end
    d["obs0"] = np.stack([c["obs0"] for c in cases])
    d["actions"] = np.stack([c["actions"] for c in cases])
    d["steps"] = np.stack([c["steps"] for c in cases])         # [case, step, 7 + 12]
    d["cwnd"] = np.stack([c["cwnd"] for c in cases])           # [case, step] window after the step's action
    n_feat_hist = cases[0]["obs"].shape[1]
    d["obs_tail"] = np.stack([c["obs"][:, n_feat_hist - c["n_features"]:] for c in cases])
    d["obs_full"] = np.stack([c["obs"] for c in cases[:keep_full_obs]])
    d["done"] = np.stack([c["done"] for c in cases])
    d["columns"] = np.array(["sent", "acked", "lost", "rate", "cur_time", "run_dur",
                             "reward"] + METRICS)
    return d


def uniform_pm1(rs, n):
    return rs.uniform(-1.0, 1.0, n)


def uniform_0_2(rs, n):
    return rs.uniform(0.0, 2.0, n)


def uniform_big(rs, n):
    # large swings, both signs: exercises the MIN_RATE/MAX_RATE clamps
    return rs.uniform(-30.0, 30.0, n)


def uniform_pm1_pairs(rs, n):
    # [rate action, cwnd action]: the window stays near its start of 25 packets, below most BDPs
    return rs.uniform(-1.0, 1.0, (n, 2))


def rate_pm1_cwnd_up(rs, n):
    # the window grows ~5 %/step from 25: the env moves from window-limited to rate-limited
    return np.stack([rs.uniform(-1.0, 1.0, n), rs.uniform(0.0, 4.0, n)], axis=1)


def gen_single(ns, name, seeds, action_fn, n_steps=400, **kw):
    cases = []
    n_features = len(kw.get("features", DEFAULT_FEATURES).split(","))
    n_episodes = kw.get("n_episodes", 1)
    for seed in seeds:
        with contextlib.redirect_stdout(io.StringIO()):
            eps = run_env_episodes(ns, seed, n_steps, action_fn, **kw)
        for e, ep in enumerate(eps):
            ep["seed"], ep["episode"], ep["n_features"] = seed, e, n_features
            cases.append(ep)
    d = pack(cases)
    d["history_len"] = np.int64(kw.get("history_len", 10))
    d["features"] = np.array(kw.get("features", DEFAULT_FEATURES).split(","))
    if kw.get("fixed") is not None:
        d["fixed"] = np.array(kw["fixed"], dtype=np.float64)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-28s %3d episodes x %d steps  -> %s (%.0f KiB)" % (
        name, len(cases), n_steps, os.path.basename(path), os.path.getsize(path) / 1024.0))


def gen_two_sender(ns, name, seeds, n_steps=200, cwnd=False, noise=False, features=None, history_len=10):
    """Engine-level two-sender cases.

    SimulatedNetworkEnv never builds a second sender, so the harness drives the
    reference Link/Sender/Network classes directly with the env's own step
    protocol applied to both senders, and adds the lower-id-first tie-break
    (Sender.__lt__) the engine needs when two senders' events collide.

    cwnd / noise switch the engine's dormant USE_CWND / USE_LATENCY_NOISE module flags on (ns:51-54: they are read by
    Network.run_for_dur for whatever senders it holds); with cwnd every sender gets its own [rate action, cwnd action]
    per step, applied like the env applies sender 0's (ns:412-414).

    features / history_len (round 5): another observation shape for BOTH senders (the default three features and ten intervals
    otherwise); such a file also keeps every sender's whole observation of every step (obs_full) and names its shape.
    """
    ns.Sender.__lt__ = lambda a, b: a.id < b.id
    ns.USE_CWND = bool(cwnd)
    ns.USE_LATENCY_NOISE = bool(noise)
    feats = (features or DEFAULT_FEATURES).split(",")
    nf = len(feats)
    all_cases = []
    for seed in seeds:
        rng = CountingRandom(seed)
        ns.random = rng
        bw = rng.uniform(100, 500)
        lat = rng.uniform(0.05, 0.5)
        queue = 1 + int(np.exp(rng.uniform(0, 8)))
        loss = rng.uniform(0.0, 0.05)
        r0 = rng.uniform(0.3, 1.5) * bw * 0.5
        r1 = rng.uniform(0.3, 1.5) * bw * 0.5
        links = [ns.Link(bw, lat, queue, loss), ns.Link(bw, lat, queue, loss)]
        senders = [ns.Sender(r0, [links[0], links[1]], 0, feats, history_len=history_len),
                   ns.Sender(r1, [links[0], links[1]], 0, feats, history_len=history_len)]
        net = ns.Network(senders, links)
        run_dur = 3 * lat
        net.run_for_dur(run_dur)
        net.run_for_dur(run_dur)
        warm = [net.cur_time, len(net.q)]
        # the env's reset() ends with the observation of every sender (ns:484): the metrics of the history's empty intervals are
        # evaluated (and cached, so:44-53) HERE, before the connection has a latency minimum -- not at the first step's get_obs()
        for s_ in senders:
            s_.get_obs()
        rs = np.random.RandomState(seed)
        actions = rs.uniform(-1.0, 1.0, (n_steps, 2))
        cwnd_actions = rs.uniform(-1.0, 3.0, (n_steps, 2)) if cwnd else np.zeros((n_steps, 2))
        rows = [[], []]
        obs_tail = [[], []]
        obs_full = [[], []]
        cwnds = []
        for t in range(n_steps):
            for i in range(2):
                senders[i].apply_rate_delta(actions[t, i])
                if cwnd:
                    senders[i].apply_cwnd_delta(cwnd_actions[t, i])
            cwnds.append([senders[0].cwnd, senders[1].cwnd])
            net.run_for_dur(run_dur)
            for s in senders:
                s.record_run()
            mis = [s.get_run_data() for s in senders]
            for i in range(2):
                mi = mis[i]
                thr, la, lo = mi.get("recv rate"), mi.get("avg latency"), mi.get("loss ratio")
                reward = (10.0 * thr / (8 * ns.BYTES_PER_PACKET) - 1e3 * la - 2e3 * lo) * ns.REWARD_SCALE
                vals = [mi.get(m) for m in METRICS]
                obs = senders[i].get_obs()
                obs_tail[i].append(np.asarray(obs, dtype=np.float64)[-nf:])
                obs_full[i].append(np.asarray(obs, dtype=np.float64))
                rows[i].append([senders[i].sent, senders[i].acked, senders[i].lost,
                                senders[i].rate, net.cur_time, 0.0, reward] + [float(v) for v in vals])
            lat0 = mis[0].get("avg latency")
            if lat0 > 0.0:
                run_dur = 0.5 * lat0
            for i in range(2):
                rows[i][-1][5] = run_dur
        all_cases.append(dict(seed=seed, params=[bw, lat, queue, loss, r0, r1, 3 * lat], warm=warm,
                              rng=[6, rng.n],
                              actions=actions, cwnd_actions=cwnd_actions, cwnd=np.array(cwnds, dtype=np.int64),
                              steps=np.array(rows, dtype=np.float64),
                              obs_tail=np.array(obs_tail, dtype=np.float64), obs_full=np.array(obs_full, dtype=np.float64)))
    d = dict(seed=np.array([c["seed"] for c in all_cases], dtype=np.int64),
             params=np.array([c["params"] for c in all_cases], dtype=np.float64),
             warm=np.array([c["warm"] for c in all_cases], dtype=np.float64),
             rng=np.array([c["rng"] for c in all_cases], dtype=np.int64),
             actions=np.stack([c["actions"] for c in all_cases]),
             cwnd_actions=np.stack([c["cwnd_actions"] for c in all_cases]),   # [case, step, sender] (zeros without cwnd)
             cwnd=np.stack([c["cwnd"] for c in all_cases]),            # [case, step, sender] window after the step's action
             steps=np.stack([c["steps"] for c in all_cases]),          # [case, sender, step, 19]
             obs_tail=np.stack([c["obs_tail"] for c in all_cases]),    # [case, sender, step, 3]
             columns=np.array(["sent", "acked", "lost", "rate", "cur_time", "run_dur",
                               "reward"] + METRICS))
    if features is not None or history_len != 10:
        d["obs_full"] = np.stack([c["obs_full"] for c in all_cases])   # [case, sender, step, history_len * features]
        d["features"] = np.array(feats)
        d["history_len"] = np.int64(history_len)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **d)
    print("%-28s %3d cases x %d steps  -> %s (%.0f KiB)" % (
        name, len(all_cases), n_steps, os.path.basename(path), os.path.getsize(path) / 1024.0))
    del ns.Sender.__lt__
    ns.USE_CWND = False
    ns.USE_LATENCY_NOISE = False


def main():
    only = sys.argv[1] if len(sys.argv) > 1 else ""   # optional: regenerate only the fixtures whose name starts with this
    ns = import_reference()
    real_gen_single, real_gen_two = globals()["gen_single"], globals()["gen_two_sender"]

    def gen_single(ns_, name, *a, **kw):
        if name.startswith(only):
            real_gen_single(ns_, name, *a, **kw)

    def gen_two_sender(ns_, name, *a, **kw):
        if name.startswith(only):
            real_gen_two(ns_, name, *a, **kw)
    cwd = os.getcwd()
    tmp = tempfile.mkdtemp(prefix="pcc_golden_")
    os.chdir(tmp)   # the reference env dumps pcc_env_log_run_N.json into the CWD
    try:
        gen_single(ns, "default_pm1", range(0, 32), uniform_pm1)
        gen_single(ns, "saturating_0_2", range(100, 108), uniform_0_2)
        gen_single(ns, "clamp_pm30", range(150, 154), uniform_big, n_steps=200)
        gen_single(ns, "allfeat_h3", range(200, 204), uniform_pm1, history_len=3,
                   features=",".join(METRICS))
        gen_single(ns, "two_episodes", [300, 301], uniform_pm1, n_episodes=2)
        gen_single(ns, "fixed_cfg2", [0, 1], uniform_pm1, fixed=(200, 0.03, 5, 0.0, 60.0))
        gen_single(ns, "fixed_q1", [2], uniform_pm1, fixed=(150, 0.05, 1, 0.0, 300.0), n_steps=200)
        gen_single(ns, "fixed_lossy", [3], uniform_pm1, fixed=(300, 0.1, 50, 0.5, 400.0), n_steps=200)
        gen_single(ns, "fixed_deepq", [4], uniform_0_2, fixed=(100, 0.05, 2981, 0.0, 150.0), n_steps=200)
        gen_two_sender(ns, "two_sender", range(400, 406))
        gen_single(ns, "cwnd_pm1", range(500, 508), uniform_pm1_pairs, cwnd=True)
        gen_single(ns, "cwnd_grow", range(520, 524), rate_pm1_cwnd_up, n_steps=200, cwnd=True)
        gen_single(ns, "cwnd_fixed_deepq", [5], rate_pm1_cwnd_up, fixed=(100, 0.05, 2981, 0.0, 150.0), n_steps=200,
                   cwnd=True)
        gen_single(ns, "noise_pm1", range(600, 608), uniform_pm1, noise=True)
        gen_single(ns, "noise_fixed_q1", [6], uniform_pm1, fixed=(150, 0.05, 1, 0.0, 300.0), n_steps=200, noise=True)
        gen_single(ns, "noise_fixed_lossy", [7], uniform_pm1, fixed=(300, 0.1, 50, 0.5, 400.0), n_steps=200, noise=True)
        gen_single(ns, "noise_fixed_deepq", [8], uniform_0_2, fixed=(100, 0.05, 2981, 0.0, 150.0), n_steps=200,
                   noise=True)
        # both dormant options at once (module globals: they apply together like they apply alone)
        gen_single(ns, "cwnd_noise_pm1", range(700, 706), uniform_pm1_pairs, cwnd=True, noise=True)
        gen_single(ns, "cwnd_noise_grow", range(720, 723), rate_pm1_cwnd_up, n_steps=200, cwnd=True, noise=True)
        # ... and with two senders on the bottleneck (the flags apply to whatever senders the engine holds)
        gen_two_sender(ns, "two_sender_cwnd", range(800, 804), cwnd=True)
        gen_two_sender(ns, "two_sender_noise", range(820, 824), noise=True)
        gen_two_sender(ns, "two_sender_cwnd_noise", range(840, 843), cwnd=True, noise=True)
        # two senders with another observation shape: all 12 features, three intervals of history (both senders)
        gen_two_sender(ns, "two_sender_allfeat_h3", range(860, 864), features=",".join(METRICS), history_len=3)
    finally:
        os.chdir(cwd)


if __name__ == "__main__":
    main()
