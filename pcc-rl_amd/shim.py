"""Real-network side of the reference: the wire format between the
PCC-Uspace plugin and an agent process, a history of monitor intervals built from its samples, and
the plugin-side rate controller driven by an exported policy.  No GPU work here -- this is what lets a
policy trained on the batched simulator serve a real sender through the reference's own plugin API.

Reference pieces mirrored (paths relative to the reference root):
  * src/udt-plugins/training/shim.py:31-42  -- the sample line the plugin sends:
        "%d;%d;%d;%d;%f;%f;%f;%f;%s;%d;%f\\n" = flow id; bytes sent; acked; lost; send start; send end;
        recv start; recv end; str(list of RTT samples); packet size; utility
  * src/gym/online/shim_env.py:104-122       -- how the agent side parses it (last complete line,
        ast.literal_eval of the RTT list) and that it answers with str(rate)
  * src/udt-plugins/training/shim.py:15-20,73-75 -- the plugin reads the rate back with float() and hands
        PCC rate * 1e6
  * src/common/sender_obs.py:20-73, 110-191  -- SenderMonitorInterval / SenderHistory and the 12 metrics
  * src/udt-plugins/testing/loaded_client.py:53-173 -- init / get_rate / give_sample / reset with a loaded model
"""
import ast

import numpy as np

from .metrics import DEFAULT_FEATURES, METRIC_NAMES, metric_info

SAMPLE_FIELDS = ("flow_id", "bytes_sent", "bytes_acked", "bytes_lost", "send_start", "send_end", "recv_start",
                 "recv_end", "rtt_samples", "packet_size", "utility")


def encode_sample(flow_id, bytes_sent, bytes_acked, bytes_lost, send_start, send_end, recv_start, recv_end,
                  rtt_samples, packet_size, utility):
    """One monitor-interval sample as the plugin puts it on the wire (training/shim.py:31-42)."""
    return ("%d;%d;%d;%d;%f;%f;%f;%f;%s;%d;%f\n" % (
        flow_id, bytes_sent, bytes_acked, bytes_lost, send_start, send_end, recv_start, recv_end,
        list(rtt_samples) if not isinstance(rtt_samples, str) else rtt_samples, packet_size, utility)).encode()


def decode_sample(data):
    """Parse what the agent side receives (online/shim_env.py:104-122): the LAST complete line of the
    buffer.  Returns a dict with SAMPLE_FIELDS."""
    text = data.decode() if isinstance(data, (bytes, bytearray)) else data
    lines = text.split("\n")
    if len(lines) < 2:
        raise ValueError("no complete sample line in %r" % (text[:60],))
    vals = lines[-2].split(";")
    if len(vals) != 11:
        raise ValueError("a sample line has 11 ';'-separated fields, got %d" % len(vals))
    return {"flow_id": int(vals[0]), "bytes_sent": int(vals[1]), "bytes_acked": int(vals[2]),
            "bytes_lost": int(vals[3]), "send_start": float(vals[4]), "send_end": float(vals[5]),
            "recv_start": float(vals[6]), "recv_end": float(vals[7]),
            "rtt_samples": [float(r) for r in ast.literal_eval(vals[8])],
            "packet_size": int(vals[9]), "utility": float(vals[10])}


def encode_rate(rate):
    """The agent's answer: the new rate as text (online/shim_env.py:103)."""
    return str(float(rate)).encode()


def decode_rate(data):
    """What the plugin reads back (training/shim.py:19); PCC gets this value * 1e6 (ibid. :75)."""
    return float(data.decode() if isinstance(data, (bytes, bytearray)) else data)


class MonitorInterval(object):
    """SenderMonitorInterval (so:20-54) for one decoded sample; `get` evaluates the 12 metrics of
    so:110-191 on the host.  conn_min is the per-connection minimum of so:158-176 carried by the history."""

    def __init__(self, sample, conn_min=None):
        self.s = sample
        self.conn_min = conn_min          # None = no entry for this sender yet
        self._cache = {}

    def get(self, name):
        if name not in self._cache:
            self._cache[name] = self._eval(name)
        return self._cache[name]

    def _eval(self, name):
        s, rtt = self.s, self.s["rtt_samples"]
        if name == "send dur":
            return s["send_end"] - s["send_start"]
        if name == "recv dur":
            return s["recv_end"] - s["recv_start"]
        if name == "send rate":
            dur = self.get("send dur")
            return 8.0 * s["bytes_sent"] / dur if dur > 0.0 else 0.0
        if name == "recv rate":
            dur = self.get("recv dur")
            return 8.0 * (s["bytes_acked"] - s["packet_size"]) / dur if dur > 0.0 else 0.0
        if name == "avg latency":
            return float(np.mean(rtt)) if len(rtt) > 0 else 0.0
        if name == "loss ratio":
            tot = s["bytes_lost"] + s["bytes_acked"]
            return s["bytes_lost"] / tot if tot > 0 else 0.0
        if name == "latency increase":
            half = int(len(rtt) / 2)
            return float(np.mean(rtt[half:]) - np.mean(rtt[:half])) if half >= 1 else 0.0
        if name in ("ack latency inflation", "sent latency inflation"):
            dur = self.get("recv dur" if name.startswith("ack") else "send dur")
            return self.get("latency increase") / dur if dur > 0.0 else 0.0
        if name == "conn min latency":
            lat = self.get("avg latency")
            if self.conn_min is not None:
                if lat != 0.0 and lat < self.conn_min:
                    self.conn_min = lat
                return self.conn_min
            if lat > 0.0:
                self.conn_min = lat
                return lat
            return 0.0
        if name == "send ratio":
            thpt, rate = self.get("recv rate"), self.get("send rate")
            return rate / thpt if (thpt > 0.0 and rate < 1000.0 * thpt) else 1.0
        if name == "latency ratio":
            mn, cur = self.get("conn min latency"), self.get("avg latency")
            return cur / mn if mn > 0.0 else 1.0
        raise KeyError("unknown monitor-interval metric %r (known: %s)" % (name, ", ".join(METRIC_NAMES)))

    def as_array(self, features):
        return np.array([self.get(f) / metric_info(f)[2] for f in features])


_EMPTY = {"flow_id": 0, "bytes_sent": 0, "bytes_acked": 0, "bytes_lost": 0, "send_start": 0.0, "send_end": 0.0,
          "recv_start": 0.0, "recv_end": 0.0, "rtt_samples": [], "packet_size": 1500, "utility": 0.0}


class SampleHistory(object):
    """SenderHistory (so:56-73) over decoded samples: the last `length` monitor intervals, oldest
    first; `as_array` is the observation the policy was trained on."""

    def __init__(self, length=10, features=DEFAULT_FEATURES, conn_min=None):
        """conn_min: the connection's minimum of per-interval mean latencies so far.  The reference keeps it in a
        module-level table keyed by flow id (so:158) that a history reset does not clear, so a history created for
        a flow that already has one starts with empty intervals evaluated against it ("latency ratio" 0/min = 0.0,
        not the 1.0 of a first history) -- the empty intervals are evaluated here, before any sample, which is the
        value the reference's lazy evaluation (oldest interval first) arrives at."""
        self.features = features.split(",") if isinstance(features, str) else list(features)
        self.conn_min = conn_min
        self.values = [MonitorInterval(dict(_EMPTY), conn_min) for _ in range(length)]
        for mi in self.values:
            mi.as_array(self.features)

    def step(self, sample):
        mi = MonitorInterval(sample, self.conn_min)
        mi.as_array(self.features)       # evaluate now, in arrival order, like the lazy cache of the reference ends up doing
        mi.get("conn min latency")
        self.conn_min = mi.conn_min
        self.values.pop(0)
        self.values.append(mi)

    def as_array(self):
        return np.array([mi.as_array(self.features) for mi in self.values]).flatten()


def apply_rate_delta(rate, action, delta_scale=0.05, min_rate=0.5, max_rate=300.0):
    """loaded_client.py:139-160 (note its own constants: DELTA_SCALE 0.05, rates in Mbps 0.5..300;
    a zero action leaves the rate alone)."""
    delta = float(action) * delta_scale
    if delta > 0:
        rate *= (1.0 + delta)
    elif delta < 0:
        rate /= (1.0 - delta)
    return min(max(rate, min_rate), max_rate)


class PolicyRateController(object):
    """The plugin-side driver of loaded_client.py:53-137 for ONE flow, with the policy behind a callable
    `act(obs) -> action` (e.g. pcc_rl_amd.export.load_policy(...)): get_rate() / give_sample(...) / reset()
    are what PCC-Uspace's Python plugin hooks call; pcc_rl_amd.udt_plugin holds the module-level
    init/get_rate/give_sample/reset keyed by flow id that the plugin loader binds."""

    def __init__(self, act, history_len=10, features=DEFAULT_FEATURES, start_rate=6.0, delta_scale=0.05,
                 min_rate=0.5, max_rate=300.0, conn_min=None):
        """conn_min: the latency minimum a flow of this id already has (so:158: the reference's table of connection
        minima is keyed by flow id and outlives the driver object; udt_plugin.init passes it on re-initialisation)."""
        self.act, self.history_len, self.features = act, history_len, features
        self.start_rate, self.delta_scale, self.min_rate, self.max_rate = start_rate, delta_scale, min_rate, max_rate
        self.rate = self.start_rate
        self.history = SampleHistory(self.history_len, self.features, conn_min=conn_min)
        self.got_data = False

    def reset(self):
        """loaded_client.py:94-110: a fresh history; no rate is sent until a new sample arrives.  Two things survive,
        as in the reference: the flow's latency minimum (so:158-176: the table of connection minima is never cleared)
        and the RATE -- reset_rate() there draws a new `current_rate` that nothing reads, so get_rate() goes on from
        the rate the flow had (pinned by tests/golden/udt_plugin.npz)."""
        self.history = SampleHistory(self.history_len, self.features, conn_min=self.history.conn_min)
        self.got_data = False

    def give_sample(self, bytes_sent, bytes_acked, bytes_lost, send_start, send_end, recv_start, recv_end,
                    rtt_samples, packet_size, utility, flow_id=0):
        self.history.step({"flow_id": flow_id, "bytes_sent": bytes_sent, "bytes_acked": bytes_acked,
                           "bytes_lost": bytes_lost, "send_start": send_start, "send_end": send_end,
                           "recv_start": recv_start, "recv_end": recv_end, "rtt_samples": list(rtt_samples),
                           "packet_size": packet_size, "utility": utility})
        self.got_data = True

    def get_rate(self):
        if self.got_data:
            # (the observation goes to the agent as the float64 array the history holds, loaded_client.py:78; an exported
            # policy casts it itself)
            action = float(np.asarray(self.act(self.history.as_array())).reshape(-1)[0])
            self.rate = apply_rate_delta(self.rate, action, self.delta_scale, self.min_rate, self.max_rate)
        return self.rate * 1e6
