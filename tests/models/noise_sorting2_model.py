"""tests/models/noise_sorting_model.py for ONE OR TWO senders on the link (the reference's two-sender envs share one queue and one
random stream): the same restatement of a monitor interval with USE_LATENCY_NOISE -- counts, two sorts and a scan, no heap --
checked bit for bit against the oracle's event loop (tests/test_noise_formulation.py).  TEST INFRASTRUCTURE: the design
study for the two-sender version of pcc-rl_amd/csrc/pcc_noise_sorted.hip, which is not built (two senders with latency noise
still run the event loop on one lane per env).

What two senders change: the reference orders events as (time, sender, 'A' < 'S', hop, latency, dropped) -- the sender id
comes BEFORE the kind of event -- so
  * the SENDs of both senders are one sequence merged by (time, sender); the link's queue recurrence scans it in that order;
  * a hop-1 arrival (a, s') comes before the SEND (t, s) iff a < t, or a == t and s' <= s; a SEND (t, s) before the arrival
    (a, s') iff t < a, or t == a and s < s';
  * the draw index of an event = 2 x (SENDs of either sender before it) + (hop-1 arrivals of either sender before it);
  * hop-1 and hop-2 arrivals sort by (time, sender, latency, dropped); acknowledgements and RTT lists are per sender, in that order;
  * blocks of SENDs are cut by TIME: everything sent less than ~dl after the block's first SEND (no arrival of the block can
    come before a SEND of the block).
"""
import random

import numpy as np

from oracle.pcc_oracle_py import PyOracleEnv, SCALES


class SortedNoiseEnvS(PyOracleEnv):
    """PyOracleEnv with latency noise, one or two senders, no window: the interval by sorting, the rest inherited."""

    TABLE = 1 << 21

    def __init__(self, seed=0, latency_noise=1.1, **kw):
        super().__init__(seed=seed, latency_noise=latency_noise, **kw)
        assert not self.use_cwnd and self.latency_noise
        self.block_sizes = []
        self.used = 0

    def reset(self):
        for _ in range(self.used):
            self.rng.random()
        self.used = 0
        self._new_params()
        twin = random.Random()
        twin.setstate(self.rng.getstate())
        self.table = np.array([twin.random() for _ in range(self.TABLE)])
        self.now = 0.0
        self.p1, self.p2 = [], []      # arrivals not yet processed: (time, sender, latency, dropped), in key order
        self.nsend = [1.0 / r for r in self.rate]
        self.minlat = [None] * self.S
        empty = [0.0] * 12
        empty[10] = empty[11] = 1.0
        row = [empty[f] / SCALES[f] for f in self.fids]
        self.hist = [[list(row) for _ in range(self.H)] for _ in range(self.S)]
        self.run_dur = 3 * self.dl
        self.steps = 0
        self._mi(self.run_dur)
        self._mi(self.run_dur)
        return self._obs()

    def _u(self, idx):
        return self.table[self.used + idx]

    def _send(self, t, s, u_noise, u_loss):
        qd = max(0.0, self.q - (t - self.tq))
        ll = self.dl + qd
        ll *= 1.0 + (self.latency_noise - 1.0) * u_noise
        if u_loss < self.lr:
            ok = False
        else:
            self.q, self.tq = qd, t
            extra = 1.0 / self.bw
            if extra + self.q > self.maxq:
                ok = False
            else:
                self.q += extra
                ok = True
        return (t + ll, s, 0.0 + ll, not ok)

    @staticmethod
    def _arrivals_before_send(times_by_sender, t, s):
        """how many hop-1 arrivals (sorted times per sender) come before the SEND (t, s)"""
        c = 0
        for sp, arr in enumerate(times_by_sender):
            c += int(np.searchsorted(arr, t, side="right" if sp <= s else "left"))
        return c

    def _sends_before_arrival(self, sends, a, sp):
        """how many of the interval's SENDs (merged, (t, s) sorted) come before the hop-1 arrival (a, sp)"""
        n = 0
        for s in range(self.S):
            n += int(np.searchsorted(sends[s], a, side="left" if s >= sp else "right"))
        return n

    def _mi(self, dur):
        S = self.S
        end = self.now + dur
        dl, noise = self.dl, self.latency_noise
        gap = [1.0 / r for r in self.rate]
        self.t0 = self.now
        # ---- SEND times before `end`, per sender (the reference's own additions), merged by (time, sender)
        ts, t_next = [], []
        for s in range(S):
            lst, t = [], self.nsend[s]
            while t < end:
                lst.append(t)
                t = t + gap[s]
            ts.append(np.array(lst))
            t_next.append(t)
        merged = sorted((float(t), s) for s in range(S) for t in ts[s])
        K = len(merged)
        old_by_sender = [np.array([e[0] for e in self.p1 if e[1] == s]) for s in range(S)]
        new_by_sender = [np.empty(0) for _ in range(S)]
        new1 = []
        k0 = 0
        while k0 < K:
            t_first = merged[k0][0]
            k1 = k0
            while k1 < K and merged[k1][0] < t_first + 0.9 * dl:
                k1 += 1
            cs = [self._arrivals_before_send(old_by_sender, t, s) + self._arrivals_before_send(new_by_sender, t, s)
                  for (t, s) in merged[k0:k1]]
            blk = []
            for j, k in enumerate(range(k0, k1)):          # (the queue recurrence: a scan over the merged SENDs)
                t, s = merged[k]
                idx = 2 * k + cs[j]
                blk.append(self._send(t, s, self._u(idx), self._u(idx + 1)))
            assert min(e[0] for e in blk) > merged[k1 - 1][0], "a block's own arrivals must lie behind its last SEND"
            new1 += blk
            new_by_sender = [np.sort(np.concatenate([new_by_sender[s], [e[0] for e in blk if e[1] == s]])) for s in range(S)]
            self.block_sizes.append(k1 - k0)
            k0 = k1
        # ---- hop-1 arrivals: one sort by (time, sender, latency, dropped); those before `end` are processed
        a1 = sorted(self.p1 + new1)
        n1 = 0
        new2 = []
        while n1 < len(a1) and a1[n1][0] < end:
            a, sp, lat, dropped = a1[n1]
            idx = 2 * self._sends_before_arrival(ts, a, sp) + n1
            ll = dl + max(0.0, 0.0 - (a - 0.0))
            ll *= 1.0 + (noise - 1.0) * self._u(idx)
            new2.append((a + ll, sp, lat + ll, dropped))
            n1 += 1
        # ---- hop-2 arrivals: one more sort; those before `end`: acknowledgements and loss reports, per sender
        a2 = sorted(self.p2 + new2)
        n2 = 0
        acked, lost, rtts = [0] * S, [0] * S, [[] for _ in range(S)]
        while n2 < len(a2) and a2[n2][0] < end:
            b, sp, lat, dropped = a2[n2]
            if dropped:
                lost[sp] += 1
            else:
                acked[sp] += 1
                rtts[sp].append(lat)
            n2 += 1
        # ---- the event that ends the interval: (time, sender, 'A' < 'S', hop, latency, dropped)
        cands = [((t_next[s], s, 1, 0, 0.0, False), "send") for s in range(S)]
        if n1 < len(a1):
            e = a1[n1]
            cands.append(((e[0], e[1], 0, 1, e[2], e[3]), "hop1"))
        if n2 < len(a2):
            e = a2[n2]
            cands.append(((e[0], e[1], 0, 2, e[2], e[3]), "hop2"))
        key, what = min(cands)
        sent = [len(ts[s]) for s in range(S)]
        draws = 2 * K + n1
        rest1, rest2 = a1[n1:], a2[n2:]
        if what == "send":
            s = key[1]
            idx = 2 * K + n1
            rest1 = sorted(rest1 + [self._send(key[0], s, self._u(idx), self._u(idx + 1))])
            t_next[s] = key[0] + gap[s]
            sent[s] += 1
            draws += 2
        elif what == "hop1":
            a, sp, lat, dropped = rest1.pop(0)
            ll = dl + max(0.0, 0.0 - (a - 0.0))
            ll *= 1.0 + (noise - 1.0) * self._u(2 * K + n1)
            rest2 = sorted(rest2 + [(a + ll, sp, lat + ll, dropped)])
            draws += 1
        else:
            b, sp, lat, dropped = rest2.pop(0)
            if dropped:
                lost[sp] += 1
            else:
                acked[sp] += 1
                rtts[sp].append(lat)
        self.now = key[0]
        self.nsend = t_next
        self.p1, self.p2 = rest1, rest2
        self.used += draws
        self.draws += draws
        self.sent, self.acked, self.lost, self.rtts = sent, acked, lost, rtts
