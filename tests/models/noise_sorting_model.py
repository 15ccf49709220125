"""USE_LATENCY_NOISE without a heap: a restatement of one monitor interval as counts, sorts and a scan, checked bit for bit
against the oracle (tests/test_noise_formulation.py).  TEST INFRASTRUCTURE: the design study for the next GPU formulation of
the option -- today the product runs the reference's event loop on one lane per env
(pcc-rl_amd/csrc/pcc_retire_env.h: event_engine), whose longest env's ~5 000 events, one after the other, ARE the launch.

What makes the option look sequential: every link latency is multiplied by one more draw of the env's random stream
(ns:150-151, 171-172), and the stream is consumed in EVENT order -- a SEND takes two draws (noise, loss), the arrival at the
return link (hop 1) one, the arrival at the sender (hop 2) none -- so which draw a packet gets depends on how the events of
all packets interleave, and with noise they overtake each other.

What makes it parallel after all (one sender, no window):
  * SEND times do not depend on anything drawn: t_0 = the pending SEND, t_{k+1} = t_k + 1/rate.
  * The draw index of an event inside the interval is 2 x (SENDs before it) + (hop-1 arrivals before it).
  * A hop-1 arrival is at least dl after its SEND (the noise factor is >= 1, the queue delay >= 0), so the hop-1 arrivals
    before SEND k all belong to packets sent more than dl ago: SENDs can be processed in BLOCKS of ~dl/gap packets whose
    counts "arrivals before me" depend on earlier blocks only -- a binary search each, all lanes at once.  What stays a scan
    inside a block is the link's queue recurrence (ns:66-84), the same one the product's send half already runs.
  * The hop-1 arrivals of an interval, old and new, are one sort by the reference's key (time, hop, latency, dropped); an
    arrival's draw index is 2 x (SENDs strictly before it: 'A' < 'S' at equal times) + its rank.  Its hop-2 time follows.
  * The hop-2 arrivals are one more sort; those before the end of the interval are the interval's acknowledgements and loss
    reports, in the order the reference appends their RTTs.
  * The event that ends the interval (the first at or after `end`, still processed: ns:128-131) is the smallest of three
    candidates -- the next SEND, the first hop-1 arrival, the first hop-2 arrival not yet due.
Per interval: two sorts of nearly sorted keys, K binary searches, one scan -- wavefront work for a group of lanes per env.
"""
import random

import numpy as np

from oracle.pcc_oracle_py import PyOracleEnv


class SortedNoiseEnv(PyOracleEnv):
    """PyOracleEnv with latency noise, one sender, no window: the interval by sorting, the rest inherited."""

    TABLE = 1 << 21

    def __init__(self, seed=0, latency_noise=1.1, **kw):
        super().__init__(seed=seed, latency_noise=latency_noise, **kw)
        assert self.S == 1 and not self.use_cwnd and self.latency_noise
        self.block_sizes = []          # how parallel it was: SENDs per block
        self.used = 0

    def reset(self):
        # the stream by index: the draws after the parameter draws, as a table (the oracle consumes them one by one)
        for _ in range(self.used):
            self.rng.random()
        self.used = 0
        self._new_params()
        twin = random.Random()
        twin.setstate(self.rng.getstate())
        self.table = np.array([twin.random() for _ in range(self.TABLE)])
        self.now = 0.0
        self.p1 = []                   # hop-1 arrivals not yet processed: (time, latency, dropped), in key order
        self.p2 = []                   # hop-2 arrivals not yet processed
        self.nsend = 1.0 / self.rate[0]
        self.minlat = [None]
        empty = [0.0] * 12
        empty[10] = empty[11] = 1.0
        from oracle.pcc_oracle_py import SCALES
        row = [empty[f] / SCALES[f] for f in self.fids]
        self.hist = [[list(row) for _ in range(self.H)]]
        self.run_dur = 3 * self.dl
        self.steps = 0
        self._mi(self.run_dur)
        self._mi(self.run_dur)
        return self._obs()

    def _u(self, idx):
        return self.table[self.used + idx]

    def _send(self, t, u_noise, u_loss):
        """The link's part of one SEND (ns:66-84, 170-175): returns (hop-1 arrival time, latency, dropped)."""
        qd = max(0.0, self.q - (t - self.tq))
        ll = self.dl + qd
        ll *= 1.0 + (self.latency_noise - 1.0) * u_noise          # random.uniform(1.0, MAX_LATENCY_NOISE)
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
        return (t + ll, 0.0 + ll, not ok)

    def _mi(self, dur):
        end = self.now + dur
        dl, noise = self.dl, self.latency_noise
        gap = 1.0 / self.rate[0]
        self.t0 = self.now
        # ---- SEND times before `end` (the reference's own additions)
        ts = []
        t = self.nsend
        while t < end:
            ts.append(t)
            t = t + gap
        t_next = t
        K = len(ts)
        ts_a = np.array(ts)
        old_a = np.array([e[0] for e in self.p1])
        # ---- SENDs in blocks: hop-1 arrivals before SEND k = old ones <= t_k + new ones of EARLIER BLOCKS <= t_k
        new1 = []                       # (time, latency, dropped) of this interval's packets, in send order
        new_sorted = np.empty(0)
        W = max(1, int(dl / gap) - 1)
        k0 = 0
        while k0 < K:
            k1 = min(K, k0 + W)
            tk = ts_a[k0:k1]
            c = np.searchsorted(old_a, tk, side="right") + np.searchsorted(new_sorted, tk, side="right")
            for j, k in enumerate(range(k0, k1)):            # (the queue recurrence: a scan)
                idx = 2 * k + int(c[j])
                new1.append(self._send(ts[k], self._u(idx), self._u(idx + 1)))
            blk = np.array([e[0] for e in new1[k0:k1]])
            assert blk.min() > tk[-1], "a block's own arrivals must lie behind its last SEND"
            new_sorted = np.sort(np.concatenate([new_sorted, blk]))
            self.block_sizes.append(k1 - k0)
            k0 = k1
        # ---- hop-1 arrivals: one sort by the reference's key; those before `end` are processed
        a1 = sorted(self.p1 + new1)
        n1 = 0
        new2 = []
        while n1 < len(a1) and a1[n1][0] < end:
            a, lat, dropped = a1[n1]
            idx = 2 * int(np.searchsorted(ts_a, a, side="left")) + n1     # SENDs strictly before it ('A' < 'S'), arrivals before it
            ll = dl + max(0.0, 0.0 - (a - 0.0))
            ll *= 1.0 + (noise - 1.0) * self._u(idx)
            new2.append((a + ll, lat + ll, dropped))
            n1 += 1
        # ---- hop-2 arrivals: one more sort; those before `end` are this interval's acknowledgements and loss reports
        a2 = sorted(self.p2 + new2)
        n2 = 0
        acked = lost = 0
        rtts = []
        while n2 < len(a2) and a2[n2][0] < end:
            if a2[n2][2]:
                lost += 1
            else:
                acked += 1
                rtts.append(a2[n2][1])
            n2 += 1
        # ---- the event that ends the interval: the smallest of three by (time, 'A' < 'S', hop, latency, dropped)
        cands = [((t_next, 1, 0, 0.0, False), "send")]
        if n1 < len(a1):
            cands.append(((a1[n1][0], 0, 1, a1[n1][1], a1[n1][2]), "hop1"))
        if n2 < len(a2):
            cands.append(((a2[n2][0], 0, 2, a2[n2][1], a2[n2][2]), "hop2"))
        key, what = min(cands)
        sent = K
        draws = 2 * K + n1
        rest1, rest2 = a1[n1:], a2[n2:]
        if what == "send":
            idx = 2 * K + n1
            rest1 = sorted(rest1 + [self._send(t_next, self._u(idx), self._u(idx + 1))])
            t_next = t_next + gap
            sent += 1
            draws += 2
        elif what == "hop1":
            a, lat, dropped = rest1.pop(0)
            ll = dl + max(0.0, 0.0 - (a - 0.0))
            ll *= 1.0 + (noise - 1.0) * self._u(2 * K + n1)
            rest2 = sorted(rest2 + [(a + ll, lat + ll, dropped)])
            draws += 1
        else:
            b, lat, dropped = rest2.pop(0)
            if dropped:
                lost += 1
            else:
                acked += 1
                rtts.append(lat)
        self.now = key[0]
        self.nsend = t_next
        self.p1, self.p2 = rest1, rest2
        self.used += draws
        self.draws += draws
        self.sent, self.acked, self.lost, self.rtts = [sent], [acked], [lost], [rtts]
