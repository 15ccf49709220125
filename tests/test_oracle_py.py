"""Pins the pure-Python oracle sibling (oracle/pcc_oracle_py.py, the "reference-class" CPU
baseline bench.py times) bit-for-bit against the golden vectors from the reference."""
import os

import numpy as np

from oracle.pcc_oracle_py import NAMES, PyOracleEnv

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(G, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def check(env, d, i):
    obs0 = env.reset()
    assert np.array_equal(obs0, d["obs0"][i])
    assert env.now == d["warm"][i][0] and len(env.heap) == int(d["warm"][i][1])
    nf = d["obs_tail"].shape[2]
    for t in range(d["actions"].shape[1]):
        obs, rew, done, _ = env.step(d["actions"][i, t])
        assert np.array_equal(np.array(env.last_rows[0], dtype=np.float64), d["steps"][i, t]), (i, t)
        assert np.array_equal(obs[-nf:], d["obs_tail"][i, t])
        assert done == bool(d["done"][i, t])
    assert env.draws == int(d["rng"][i][1])


def test_default_episodes():
    d = load("default_pm1")
    for i in (0, 1, 5):
        check(PyOracleEnv(seed=int(d["seed"][i])), d, i)


def test_all_features_short_history():
    d = load("allfeat_h3")
    check(PyOracleEnv(seed=int(d["seed"][0]), history_len=3, features=NAMES), d, 0)


def test_fixed_and_saturating():
    d = load("fixed_lossy")
    check(PyOracleEnv(seed=int(d["seed"][0]), fixed=tuple(d["fixed"])), d, 0)
    d = load("saturating_0_2")
    check(PyOracleEnv(seed=int(d["seed"][0])), d, 0)


def test_two_senders():
    d = load("two_sender")
    for i in (0, 1):
        bw, lat, queue, loss, r0, r1, _ = d["params"][i]
        env = PyOracleEnv(seed=int(d["seed"][i]), n_senders=2, fixed=(bw, lat, queue, loss, r0, r1), ctor_draws=6)
        env.reset()
        assert env.now == d["warm"][i][0]
        for t in range(d["actions"].shape[1]):
            obs, rew, done, _ = env.step(d["actions"][i, t])
            for s in range(2):
                assert np.array_equal(np.array(env.last_rows[s], dtype=np.float64), d["steps"][i, s, t]), (i, s, t)
                assert np.array_equal(obs[s][-3:], d["obs_tail"][i, s, t])


def test_two_senders_with_another_observation_shape():
    d = load("two_sender_allfeat_h3")
    feats, H = tuple(str(f) for f in d["features"]), int(d["history_len"])
    for i in (0, 3):
        bw, lat, queue, loss, r0, r1, _ = d["params"][i]
        env = PyOracleEnv(seed=int(d["seed"][i]), n_senders=2, fixed=(bw, lat, queue, loss, r0, r1), ctor_draws=6,
                          history_len=H, features=feats)
        env.reset()
        assert env.now == d["warm"][i][0]
        for t in range(d["actions"].shape[1]):
            obs, rew, done, _ = env.step(d["actions"][i, t])
            for s in range(2):
                assert np.array_equal(np.array(env.last_rows[s], dtype=np.float64), d["steps"][i, s, t]), (i, s, t)
                assert np.array_equal(np.asarray(obs[s], dtype=np.float64), d["obs_full"][i, s, t]), (i, s, t)


def test_use_cwnd_option():
    d = load("cwnd_pm1")
    for i in (0, 3):
        env = PyOracleEnv(seed=int(d["seed"][i]), use_cwnd=True)
        check(env, d, i)
    d = load("cwnd_grow")
    env = PyOracleEnv(seed=int(d["seed"][0]), use_cwnd=True)
    check(env, d, 0)
    assert env.cwnd[0] == int(d["cwnd"][0, -1])


def test_use_latency_noise_option():
    d = load("noise_pm1")
    for i in (0, 5):
        check(PyOracleEnv(seed=int(d["seed"][i]), latency_noise=1.1), d, i)
    d = load("noise_fixed_lossy")
    check(PyOracleEnv(seed=int(d["seed"][0]), fixed=tuple(d["fixed"]), latency_noise=1.1), d, 0)


def test_engine_options_with_two_senders():
    for name, cwnd, noise in (("two_sender_cwnd", True, False), ("two_sender_noise", False, True),
                              ("two_sender_cwnd_noise", True, True)):
        d = load(name)
        i = 1
        bw, lat, queue, loss, r0, r1, _ = d["params"][i]
        env = PyOracleEnv(seed=int(d["seed"][i]), n_senders=2, fixed=(bw, lat, queue, loss, r0, r1), ctor_draws=6,
                          use_cwnd=cwnd, latency_noise=1.1 if noise else None)
        env.reset()
        assert env.now == d["warm"][i][0]
        for t in range(d["actions"].shape[1]):
            a = np.stack([d["actions"][i, t], d["cwnd_actions"][i, t]], axis=1) if cwnd else d["actions"][i, t]
            obs, rew, done, _ = env.step(a)
            for s in range(2):
                assert np.array_equal(np.array(env.last_rows[s], dtype=np.float64), d["steps"][i, s, t]), (name, s, t)
