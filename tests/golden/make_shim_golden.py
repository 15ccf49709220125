#!/usr/bin/env python3
"""Golden vectors for pcc-rl_amd/shim.py, generated from the UNMODIFIED reference in the build
container (it cannot travel): wire lines produced by the reference plugin's own give_sample
(src/udt-plugins/training/shim.py:24-43, with a socket stand-in that records what is sent) and the
observation arrays of the reference's SenderHistory (src/common/sender_obs.py:56-73) after each sample.

    python tests/golden/make_shim_golden.py   ->  tests/golden/shim_samples.npz
"""
import contextlib
import io
import os
import sys

import numpy as np

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REF, "udt-plugins", "training"))
sys.path.insert(0, REF)
with contextlib.redirect_stdout(io.StringIO()):
    import shim                       # the reference plugin module (prints on import)
    from common import sender_obs


class Recorder(object):
    def __init__(self):
        self.sent = []

    def send(self, data):
        self.sent.append(data)


def main():
    rs = np.random.RandomState(7)
    features = ["sent latency inflation", "latency ratio", "send ratio", "recv rate", "loss ratio", "conn min latency"]
    drv = object.__new__(shim.PccShimDriver)     # no socket connect
    drv.sock, drv.replay_rate = Recorder(), True
    hist = sender_obs.SenderHistory(4, features, 12345)
    fields, lines, obs = [], [], []
    t = 0.0
    for k in range(40):
        n_rtt = int(rs.randint(0, 9)) if k % 7 else 0
        base = 0.02 + 0.2 * rs.rand()
        rtts = [float(base + 0.01 * rs.rand()) for _ in range(n_rtt)]
        sent = int(rs.randint(0, 200)) * 1500
        lost = int(rs.randint(0, 20)) * 1500
        acked = max(0, sent - lost - int(rs.randint(0, 5)) * 1500)
        dur = float(0.05 + rs.rand()) if k % 11 else 0.0
        row = (3, sent, acked, lost, t, t + dur, t + 0.03, t + 0.03 + dur, rtts, 1500, float(rs.randn()))
        t += dur
        drv.replay_rate = True
        drv.give_sample(*row)
        lines.append(drv.sock.sent[-1])
        hist.step(sender_obs.SenderMonitorInterval(12345, bytes_sent=row[1], bytes_acked=row[2], bytes_lost=row[3],
                                                   send_start=row[4], send_end=row[5], recv_start=row[6], recv_end=row[7],
                                                   rtt_samples=row[8], packet_size=row[9]))
        obs.append(hist.as_array())
        fields.append(repr(row))
    np.savez(os.path.join(HERE, "shim_samples.npz"), lines=np.array(lines), obs=np.array(obs),
             features=np.array(features), history_len=4, rows=np.array(fields))
    print("wrote shim_samples.npz:", len(lines), "samples")
    plugin_golden()


PLUGIN_W = [0.37, -0.021, 0.0113]      # the stub agent: act(obs) = tanh(sum_k obs[k] * W[k % 3] * (1 + k / 30))


def stub_act(obs):
    obs = np.asarray(obs, dtype=np.float64).reshape(-1)
    return float(np.tanh(sum(obs[k] * PLUGIN_W[k % 3] * (1.0 + k / 30.0) for k in range(obs.size))))


def plugin_golden():
    """The reference's deployment surface -- the module-level init / get_rate / give_sample / reset of
    src/udt-plugins/testing/loaded_client.py:132-173 -- driven with a stub in place of its TensorFlow agent
    (loaded_agent.LoadedModelAgent): a script of calls on two flows, including resets, and what every get_rate
    returned.  -> tests/golden/udt_plugin.npz"""
    import types
    stub = types.ModuleType("loaded_agent")

    class LoadedModelAgent(object):
        def __init__(self, path):
            self.path = path

        def act(self, ob):
            return stub_act(ob)

        def reset(self):
            pass

    stub.LoadedModelAgent = LoadedModelAgent
    sys.modules["loaded_agent"] = stub
    sys.path.insert(0, os.path.join(REF, "udt-plugins", "testing"))
    with contextlib.redirect_stdout(io.StringIO()):
        import loaded_client as lc
    rs = np.random.RandomState(11)
    script, rates, obs = [], [], []
    t = {3: 0.0, 9: 0.0}

    def sample(flow):
        n_rtt = int(rs.randint(0, 7))
        base = 0.03 + 0.1 * rs.rand()
        rtts = [float(base + 0.01 * rs.rand()) for _ in range(n_rtt)]
        sent = int(rs.randint(1, 120)) * 1500
        lost = int(rs.randint(0, 6)) * 1500
        acked = max(0, sent - lost)
        dur = float(0.05 + 0.2 * rs.rand())
        row = (flow, sent, acked, lost, t[flow], t[flow] + dur, t[flow] + 0.03, t[flow] + 0.03 + dur, rtts, 1500, 0.0)
        t[flow] += dur
        return row

    def do(op, *args):
        script.append(repr((op,) + args))
        if op == "init":
            lc.init(*args)
        elif op == "reset":
            lc.reset(*args)
        elif op == "give_sample":
            lc.give_sample(*args)
        elif op == "get_rate":
            rates.append(float(lc.get_rate(*args)))
            obs.append(np.array(lc.PccGymDriver.get_by_flow_id(args[0]).history.as_array(), dtype=np.float64))

    do("init", 3)
    do("get_rate", 3)
    for k in range(6):
        do("give_sample", *sample(3))
        do("get_rate", 3)
    do("init", 9)
    for k in range(4):
        do("give_sample", *sample(9))
        do("get_rate", 9)
        do("give_sample", *sample(3))
        do("get_rate", 3)
    do("reset", 3)
    do("get_rate", 3)
    for k in range(5):
        do("give_sample", *sample(3))
        do("get_rate", 3)
    do("reset", 9)
    do("give_sample", *sample(9))
    do("give_sample", *sample(9))
    do("get_rate", 9)
    # a flow id that is initialised AGAIN: a new driver object, but the module-level table of connection minima
    # (sender_obs.py:158) still holds the flow's entry
    do("init", 3)
    do("get_rate", 3)
    for k in range(3):
        do("give_sample", *sample(3))
        do("get_rate", 3)
    np.savez(os.path.join(HERE, "udt_plugin.npz"), script=np.array(script), rates=np.array(rates), obs=np.array(obs),
             w=np.array(PLUGIN_W))
    print("wrote udt_plugin.npz:", len(script), "calls,", len(rates), "rates")


if __name__ == "__main__":
    main()
