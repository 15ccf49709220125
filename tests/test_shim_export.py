"""The real-network pieces: the plugin <-> agent wire format and the
monitor-interval history against vectors generated from the unmodified reference
(tests/golden/make_shim_golden.py), and the policy export / plugin-side controller round trip."""
import ast
import os

import pytest

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
    assert ctl.get_rate() == r      # like the reference's reset(): the history is new, the rate stands (loaded_client.py:94-110)


def test_udt_plugin_module_equals_the_reference_loaded_client():
    """The deployment surface: module-level init / get_rate / give_sample / reset keyed by flow id
    (src/udt-plugins/testing/loaded_client.py:132-173), replayed against what the reference module returned for the
    same script of calls on two flows -- resets included -- with the same stub agent behind both."""
    from pcc_rl_amd import udt_plugin
    with np.load(os.path.join(os.path.dirname(G), "udt_plugin.npz"), allow_pickle=False) as z:
        d = {k: z[k] for k in z.files}
    w = d["w"]

    def stub_act(obs):
        obs = np.asarray(obs, dtype=np.float64).reshape(-1)
        return float(np.tanh(sum(obs[k] * w[k % 3] * (1.0 + k / 30.0) for k in range(obs.size))))

    udt_plugin.set_policy(stub_act)
    udt_plugin._flows.clear()
    k = 0
    for call in d["script"]:
        op, *args = ast.literal_eval(str(call))
        if op == "get_rate":
            got = getattr(udt_plugin, op)(*args)
            assert got == d["rates"][k], (k, call)
            assert np.array_equal(udt_plugin._flows[args[0]].history.as_array(), d["obs"][k]), (k, call)
            k += 1
        else:
            getattr(udt_plugin, op)(*args)
    assert k == len(d["rates"])


def test_export_leaves_the_training_policy_where_it_is(tmp_path):
    """export_policy works on a copy: the caller's modules keep their device and dtype (nn.Module.to() is in place)."""
    pol = MlpPolicy(30, 1).to(torch.float64)      # a dtype the export does not use stands in for "another device" on a CPU box
    before = [(p.device, p.dtype, p.data_ptr()) for p in pol.pi.parameters()]
    mod = export._Exported(__import__("copy").deepcopy(pol.pi), pol.log_std).to(torch.float32)
    assert [(p.device, p.dtype, p.data_ptr()) for p in pol.pi.parameters()] == before
    assert next(mod.pi.parameters()).dtype == torch.float32
    pol32 = MlpPolicy(30, 1)
    ptrs = [p.data_ptr() for p in pol32.pi.parameters()]
    export.export_policy(pol32, str(tmp_path))
    assert [p.data_ptr() for p in pol32.pi.parameters()] == ptrs


def test_top_level_plugin_file_loads_like_the_reference_one():
    """examples/loaded_client.py: what `-pyhelper=loaded_client -pypath=.../examples` imports -- a top-level module (no
    package context) with the four functions of src/udt-plugins/testing/loaded_client.py:132-173."""
    import importlib.util
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples", "loaded_client.py")
    spec = importlib.util.spec_from_file_location("loaded_client", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.set_policy(lambda obs: 0.5)
    mod.init(7)
    assert mod.get_rate(7) == 6.0e6                       # no sample yet: the reset rate stands
    mod.give_sample(7, 30000, 27000, 1500, 0.0, 0.1, 0.03, 0.13, [0.03, 0.031], 1500, 1.0)
    assert mod.get_rate(7) == shim.apply_rate_delta(6.0, 0.5) * 1e6
    mod.reset(7)
    assert mod.get_rate(7) == shim.apply_rate_delta(6.0, 0.5) * 1e6


@pytest.mark.gpu
def test_deployment_side_fixtures_on_the_gpu_box(tmp_path):
    """The driver runs only `-m gpu` on the GPU box: the deployment side is checked there too --
    the same fixture comparisons as above (they need no reference, only tests/golden/*.npz), and the export taken from a policy
    that lives on the GPU, as a training run leaves it."""
    test_wire_lines_equal_the_reference_plugins()
    test_decode_takes_the_last_complete_line()
    test_history_observations_equal_the_reference()
    test_udt_plugin_module_equals_the_reference_loaded_client()
    test_top_level_plugin_file_loads_like_the_reference_one()
    torch.manual_seed(0)
    pol = MlpPolicy(30, 1).to("cuda:0")
    path = export.export_policy(pol, str(tmp_path), history_len=10, features="sent latency inflation,latency ratio,send ratio")
    assert next(pol.parameters()).device.type == "cuda"          # the training policy stays where it is
    loaded = torch.jit.load(path)
    ob = torch.randn(5, 30)
    act, stoch = loaded(ob)
    with torch.no_grad():
        mean = pol.pi(ob.to("cuda:0")).cpu()
    assert torch.allclose(act, mean, atol=1e-6) and stoch.shape == act.shape == (5, 1)
