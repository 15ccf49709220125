"""Pure-Python sibling of oracle/pcc_oracle.c: the same env hot path restated with `heapq`
and `numpy.mean`, i.e. in the reference's own algorithm class and language.

TEST INFRASTRUCTURE, NOT PRODUCT.  Two uses only:
  * tests/test_oracle_py.py pins it bit-for-bit against tests/golden/*.npz;
  * bench.py's cpu_baseline leg times it on the GPU box's host cores as the
    "reference-class CPU env" figure (the reference's own files never leave the build
    container), next to the much faster C oracle.

Own code, written from the behavioural description in SURVEY.md section 3 ("ns" =
src/gym/network_sim.py, "so" = src/common/sender_obs.py of the reference): flat state in
one object, integer event codes, no per-metric registry objects.
"""
import heapq
import math
import random

import numpy as np

ACK, SEND = 0, 1                  # 'A' < 'S' in the reference's tuple order (ns:43-44)
PKT = 1500                        # ns:46
RATE_LO, RATE_HI = 40.0, 1000.0   # ns:36-37
EPISODE_STEPS = 400               # ns:41
SCALES = (1e7, 1e7, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0, 1.0)   # so:193-206
NAMES = ("send rate", "recv rate", "recv dur", "send dur", "avg latency", "loss ratio",
         "ack latency inflation", "sent latency inflation", "conn min latency", "latency increase",
         "latency ratio", "send ratio")


class PyOracleEnv(object):
    """Single env, 1 or 2 senders, reference life cycle: construct, reset(), step(a)..."""

    def __init__(self, seed=0, history_len=10, features=("sent latency inflation", "latency ratio", "send ratio"),
                 n_senders=1, delta_scale=0.025, fixed=None, ctor_draws=5, use_cwnd=False,
                 latency_noise=None):
        self.rng = random.Random(seed)
        for _ in range(ctor_draws):           # the constructor's discarded parameter draws (ns:366)
            self.rng.random()
        self.fids = [NAMES.index(f) for f in features]
        self.H, self.S, self.delta_scale, self.fixed = history_len, n_senders, delta_scale, fixed
        self.run_dur = None
        self.draws = ctor_draws
        self.use_cwnd = use_cwnd          # ns:54: window-limited sending, [rate action, cwnd action] per step
        self.latency_noise = latency_noise  # ns:51-52: None = off, else MAX_LATENCY_NOISE (1.1 in the reference)

    # ---- parameters and reset: ns:454-484
    def _new_params(self):
        if self.fixed is not None:
            bw, dl, queue, loss = self.fixed[:4]
            rates = list(self.fixed[4:4 + self.S])
        else:
            u = self.rng.uniform
            bw = u(100, 500)
            dl = u(0.05, 0.5)
            queue = 1 + int(np.exp(u(0, 8)))
            loss = u(0.0, 0.05)
            rates = [u(0.3, 1.5) * bw for _ in range(self.S)]
            self.draws += 4 + self.S
        self.bw, self.dl, self.lr = float(bw), dl, loss
        self.maxq = queue / self.bw
        self.q, self.tq = 0.0, 0.0
        self.rate = rates
        self.rate0 = list(rates)
        self.cwnd = [25] * self.S           # ns:209, 227
        self.in_flight = [0] * self.S       # bytes_in_flight / BYTES_PER_PACKET (ns:215)

    def reset(self):
        self._new_params()
        self.now = 0.0
        self.heap = []
        self.minlat = [None] * self.S
        empty = [0.0] * 12
        empty[10] = empty[11] = 1.0
        row = [empty[f] / SCALES[f] for f in self.fids]
        self.hist = [[list(row) for _ in range(self.H)] for _ in range(self.S)]
        for s in range(self.S):
            heapq.heappush(self.heap, (1.0 / self.rate[s], s, SEND, 0, 0.0, False))
        self.run_dur = 3 * self.dl
        self.steps = 0
        self._mi(self.run_dur)
        self._mi(self.run_dur)
        return self._obs()

    def _obs(self):
        out = [np.array([v for mi in h for v in mi]) for h in self.hist]
        return out[0] if self.S == 1 else np.stack(out)

    # ---- one monitor interval: ns:123-178
    def _mi(self, dur):
        end = self.now + dur
        S = self.S
        self.sent, self.acked, self.lost = [0] * S, [0] * S, [0] * S
        self.rtts = [[] for _ in range(S)]
        self.t0 = self.now
        heap, push, pop = self.heap, heapq.heappush, heapq.heappop
        dl, rnd = self.dl, self.rng.random
        while self.now < end:
            t, s, kind, hop, lat, dropped = pop(heap)
            self.now = t
            if kind == ACK:
                if hop == 2:
                    if dropped:
                        self.lost[s] += 1
                    else:
                        self.acked[s] += 1
                        self.rtts[s].append(lat)
                    self.in_flight[s] -= 1    # ns:269, 273
                else:   # return link: never queued on, so its latency is dl + max(0, 0 - t) (ns:66-70)
                    ll = dl + max(0.0, 0.0 - (t - 0.0))
                    if self.latency_noise:        # ns:150-151
                        ll *= self.rng.uniform(1.0, self.latency_noise)
                        self.draws += 1
                    push(heap, (t + ll, s, ACK, hop + 1, lat + ll, dropped))
            else:
                # ns:158-160, 251-255: a SEND the window blocks launches no packet, but the link lines
                # below (ns:170-175) still run for it: queue update and loss draw
                launched = (not self.use_cwnd) or self.in_flight[s] < self.cwnd[s]
                if launched:
                    self.sent[s] += 1
                    self.in_flight[s] += 1
                push(heap, (t + (1.0 / self.rate[s]), s, SEND, 0, 0.0, False))
                qd = max(0.0, self.q - (t - self.tq))
                ll = dl + qd
                if self.latency_noise:            # ns:171-172: drawn before the loss decision
                    ll *= self.rng.uniform(1.0, self.latency_noise)
                    self.draws += 1
                self.draws += 1
                if rnd() < self.lr:
                    ok = False
                else:
                    self.q, self.tq = qd, t
                    extra = 1.0 / self.bw
                    if extra + self.q > self.maxq:
                        ok = False
                    else:
                        self.q += extra
                        ok = True
                if launched:
                    push(heap, (t + ll, s, ACK, 1, 0.0 + ll, not ok))

    # ---- the 12 metrics of sender s for the MI just run: so:110-191
    def _metrics(self, s, update_min):
        dur = self.now - self.t0
        r = self.rtts[s]
        m = [0.0] * 12
        m[2] = m[3] = dur
        if dur > 0.0:
            m[0] = 8.0 * (self.sent[s] * PKT) / dur
            m[1] = 8.0 * (self.acked[s] * PKT - PKT) / dur
        lat = float(np.mean(r)) if r else 0.0
        m[4] = lat
        la = (self.lost[s] + self.acked[s]) * PKT
        m[5] = (self.lost[s] * PKT) / la if la > 0 else 0.0
        half = int(len(r) / 2)
        inc = float(np.mean(r[half:]) - np.mean(r[:half])) if half >= 1 else 0.0
        m[9] = inc
        m[6] = m[7] = inc / dur if dur > 0.0 else 0.0
        prev = self.minlat[s]
        if prev is None:
            cm = lat if lat > 0.0 else 0.0
            if lat > 0.0 and update_min:
                self.minlat[s] = lat
        elif lat != 0.0 and lat < prev:
            cm = lat
            if update_min:
                self.minlat[s] = lat
        else:
            cm = prev
        m[8] = cm
        m[11] = m[0] / m[1] if (m[1] > 0.0 and m[0] < 1000.0 * m[1]) else 1.0
        m[10] = lat / cm if cm > 0.0 else 1.0
        return m

    # ---- step: ns:406-444
    def step(self, action):
        if self.run_dur is None:
            raise TypeError("step() before reset()")
        if self.use_cwnd:                  # ns:412-414 (one sender): action = [rate action, cwnd action]
            a = np.asarray(action, dtype=np.float64).reshape(self.S, 2)
            for s in range(self.S):
                d = float(a[s, 1]) * self.delta_scale   # ns:243-249, 283-289
                c = int(self.cwnd[s] * (1.0 + d)) if d >= 0.0 else int(self.cwnd[s] / (1.0 - d))
                self.cwnd[s] = min(max(c, 4), 5000)
            action = a[:, 0]
        acts = [float(action)] if np.ndim(action) == 0 else [float(a) for a in action]
        for s in range(self.S):
            d = acts[s] * self.delta_scale
            r = self.rate[s] * (1.0 + d) if d >= 0.0 else self.rate[s] / (1.0 - d)
            self.rate[s] = min(max(r, RATE_LO), RATE_HI)
        self._mi(self.run_dur)
        rewards, rows = [], []
        for s in range(self.S):
            m = self._metrics(s, True)
            rewards.append((10.0 * m[1] / (8 * PKT) - 1e3 * m[4] - 2e3 * m[5]) * 0.001)
            self.hist[s].pop(0)
            self.hist[s].append([m[f] / SCALES[f] for f in self.fids])
            rows.append(m)
        self.steps += 1
        if rows[0][4] > 0.0:
            self.run_dur = 0.5 * rows[0][4]
        self.last_rows = [[self.sent[s], self.acked[s], self.lost[s], self.rate[s], self.now, self.run_dur,
                           rewards[s]] + rows[s] for s in range(self.S)]
        done = self.steps >= EPISODE_STEPS
        if self.S == 1:
            return self._obs(), rewards[0], done, {}
        return self._obs(), rewards, done, {}


def time_episodes(seed0, n_episodes, n_steps=EPISODE_STEPS):
    """Run n_episodes seeded default-parameter episodes with U(-1,1) actions; returns
    (env_steps, packets_sent, seconds).  Used by bench.py's cpu_baseline leg."""
    import time
    steps = packets = 0
    t0 = time.perf_counter()
    for k in range(n_episodes):
        env = PyOracleEnv(seed=seed0 + k)
        acts = np.random.RandomState(seed0 + k).uniform(-1, 1, n_steps)
        env.reset()
        for t in range(n_steps):
            env.step(acts[t])
            packets += env.sent[0]
        steps += n_steps
    return steps, packets, time.perf_counter() - t0
