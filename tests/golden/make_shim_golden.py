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


if __name__ == "__main__":
    main()
