"""The real-network pieces: the plugin <-> agent wire format and the
monitor-interval history against vectors generated from the unmodified reference
(tests/golden/make_shim_golden.py), and the policy export / plugin-side controller round trip."""
import ast
import os

import numpy as np
import torch

from pcc_rl_amd import export, shim
from pcc_rl_amd.ppo import MlpPolicy

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "shim_samples.npz")


def load():
    with np.load(G, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def test_wire_lines_equal_the_reference_plugins():
    d = load()
    for line, row in zip(d["lines"], d["rows"]):
        args = ast.literal_eval(str(row))
        assert shim.encode_sample(*args) == bytes(line)
        got = shim.decode_sample(bytes(line))
        # the %f fields travel with 6 decimals; integers and the RTT list are exact
        assert got["flow_id"] == args[0] and got["bytes_sent"] == args[1] and got["bytes_acked"] == args[2]
        assert got["bytes_lost"] == args[3] and got["rtt_samples"] == list(args[8]) and got["packet_size"] == args[9]
        assert abs(got["send_end"] - args[5]) < 1e-6 and abs(got["utility"] - args[10]) < 1e-6


def test_decode_takes_the_last_complete_line():
    a = shim.encode_sample(1, 3000, 1500, 0, 0.0, 0.1, 0.02, 0.12, [0.02, 0.03], 1500, 0.5)
    b = shim.encode_sample(1, 6000, 4500, 1500, 0.1, 0.2, 0.12, 0.22, [], 1500, -0.25)
    assert shim.decode_sample(a + b)["bytes_sent"] == 6000
    assert shim.decode_sample(a + b + b"1;2;3")["bytes_sent"] == 6000      # a partial third line is ignored
    assert shim.decode_rate(shim.encode_rate(123.456)) == 123.456


def test_history_observations_equal_the_reference():
    d = load()
    hist = shim.SampleHistory(int(d["history_len"]), [str(f) for f in d["features"]])
    for row, want in zip(d["rows"], d["obs"]):
        args = ast.literal_eval(str(row))
        hist.step(dict(zip(shim.SAMPLE_FIELDS, args)))
        assert np.array_equal(hist.as_array(), want)


def test_export_signature_and_controller_round_trip(tmp_path):
    torch.manual_seed(0)
    pol = MlpPolicy(30, 1)
    path = export.export_policy(pol, str(tmp_path), history_len=10, features="sent latency inflation,latency ratio,send ratio")
    mod = torch.jit.load(path)
    ob = torch.randn(5, 30)
    act, stoch = mod(ob)                                   # ob -> (act, stochastic_act): stable_solve.py:73-81
    assert torch.allclose(act, pol.pi(ob)) and act.shape == stoch.shape == (5, 1)
    assert not torch.equal(act, stoch)
    actf = export.load_policy(str(tmp_path))
    assert np.allclose(actf(ob[0].numpy()), pol.pi(ob[:1]).detach().numpy()[0])
    ctl = shim.PolicyRateController(actf, start_rate=6.0)
    assert ctl.get_rate() == 6.0e6                         # no sample yet: the rate stands (loaded_client.py:77-81)
    ctl.give_sample(30000, 27000, 1500, 0.0, 0.1, 0.03, 0.13, [0.03, 0.031, 0.032], 1500, 1.0)
    r = ctl.get_rate()
    a = float(actf(ctl.history.as_array().astype(np.float32))[0])
    assert r == shim.apply_rate_delta(6.0, a) * 1e6 and 0.5e6 <= r <= 300e6
    ctl.reset()
    assert ctl.get_rate() == 6.0e6
