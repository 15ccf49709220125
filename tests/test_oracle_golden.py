"""Pins the CPU oracle (oracle/pcc_oracle.c) against the golden vectors that
tests/golden/make_golden.py generated from the unmodified reference env
(src/gym/network_sim.py + src/common/sender_obs.py).  Bit-exact: integer
counts, clocks, rewards, all 12 metrics and the observation vector."""
import os
import random

import numpy as np
import pytest

import oracle

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    with np.load(os.path.join(G, name + ".npz")) as z:
        return {k: z[k] for k in z.files}   # NpzFile re-reads the zip on every [] access


def test_mt19937_matches_cpython_random():
    for seed in [0, 1, 1234, 2 ** 31 - 1, 2 ** 32 + 5, 2 ** 40 + 17]:
        r = random.Random(seed)
        ref = np.array([r.random() for _ in range(2000)])
        assert np.array_equal(oracle.mt_uniforms(seed, 2000), ref)
    # KATs quoted in SURVEY.md section 7.3
    assert oracle.mt_uniforms(0, 3).tolist() == [0.8444218515250481, 0.7579544029403025, 0.420571580830845]
    assert oracle.mt_uniforms(1234, 1)[0] == 0.9664535356921388
    assert np.array_equal(oracle.mt_uniforms(7, 10, skip=13), oracle.mt_uniforms(7, 23)[13:])


def test_philox4x32_10_known_answers():
    # Random123 kat_vectors, philox4x32-10
    kat = [([0, 0, 0, 0], [0, 0], [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]),
           ([0xffffffff] * 4, [0xffffffff] * 2, [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]),
           ([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0],
            [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1])]
    for ctr, key, out in kat:
        assert oracle.philox4x32(ctr, key).tolist() == out


def test_np_mean_restatement_is_bit_exact():
    rs = np.random.RandomState(3)
    for n in list(range(1, 400)) + [1000, 4097, 8191, 8192, 8193, 10000, 16385, 30001]:
        a = rs.uniform(0.05, 30.0, n)
        assert oracle.np_mean(a) == np.mean(list(a)), n


def run_case_mt(d, i, history_len, features, n_episodes_before=0):
    """Replay golden case i from its MT seed with the reference's life cycle."""
    seed = int(d["seed"][i])
    env = oracle.OracleEnv(1, history_len, features)
    env.rng_mt(seed, skip=5)  # the constructor's five parameter draws (ns:366)
    return env


def check_episode(env, d, i, fixed=None):
    if fixed is not None:
        env.set_params(*fixed)
    obs0 = env.reset()
    assert np.array_equal(obs0, d["obs0"][i])
    p = env.params()
    gp = d["params"][i]
    assert p[0] == gp[0] and p[1] == gp[1] and p[3] == gp[3] and p[4] == gp[4] and p[5] == gp[5]
    assert round(p[2]) == round(gp[2])
    assert env.cur_time == d["warm"][i][0]
    assert env.heap_len == int(d["warm"][i][1])
    T = d["actions"].shape[1]
    HF = obs0.size
    nf = d["obs_tail"].shape[2]
    cwnd = d["actions"].ndim == 3   # [rate action, cwnd action] pairs: the USE_CWND fixtures
    for t in range(T):
        obs, rew, done, info = env.step(d["actions"][i, t] if cwnd else [d["actions"][i, t]])
        if cwnd:
            assert env.cwnd() == int(d["cwnd"][i, t])
        row = env.last_row[0]
        g = d["steps"][i, t]
        assert np.array_equal(row, g), (i, t, row - g)
        assert rew == g[6]
        assert np.array_equal(obs[HF - nf:], d["obs_tail"][i, t])
        if i < d["obs_full"].shape[0]:
            assert np.array_equal(obs, d["obs_full"][i, t])
        assert done == bool(d["done"][i, t])


@pytest.mark.parametrize("name", ["default_pm1", "saturating_0_2", "clamp_pm30", "allfeat_h3"])
def test_random_param_episodes_bit_exact(name):
    d = load(name)
    H = int(d["history_len"])
    feats = [str(f) for f in d["features"]]
    for i in range(d["seed"].shape[0]):
        env = run_case_mt(d, i, H, feats)
        check_episode(env, d, i)
        # stream position: every draw the reference made was made here too
        assert env.rng_draws == int(d["rng"][i][1])
        env.close()


def test_two_consecutive_episodes_share_the_stream():
    d = load("two_episodes")
    feats = [str(f) for f in d["features"]]
    i = 0
    while i < d["seed"].shape[0]:
        env = run_case_mt(d, i, int(d["history_len"]), feats)
        check_episode(env, d, i)
        assert int(d["episode"][i + 1]) == 1
        check_episode(env, d, i + 1)
        assert env.rng_draws == int(d["rng"][i + 1][1])
        env.close()
        i += 2


@pytest.mark.parametrize("name", ["fixed_cfg2", "fixed_q1", "fixed_lossy", "fixed_deepq"])
def test_fixed_param_episodes_bit_exact(name):
    d = load(name)
    feats = [str(f) for f in d["features"]]
    bw, dl, queue, loss, rate0 = d["fixed"]
    for i in range(d["seed"].shape[0]):
        env = run_case_mt(d, i, int(d["history_len"]), feats)
        check_episode(env, d, i, fixed=(bw, dl, queue, loss, rate0))
        assert env.rng_draws == int(d["rng"][i][1])
        env.close()


@pytest.mark.parametrize("name", ["cwnd_pm1", "cwnd_grow", "cwnd_fixed_deepq"])
def test_use_cwnd_engine_option_bit_exact(name):
    """The reference's dormant USE_CWND option (ns:54): window-limited sending incl. its quirk that
    a SEND blocked by the window still passes through the link's queue and RNG (ns:158-175)."""
    d = load(name)
    feats = [str(f) for f in d["features"]]
    fixed = tuple(d["fixed"]) if "fixed" in d else None
    for i in range(d["seed"].shape[0]):
        env = run_case_mt(d, i, int(d["history_len"]), feats)
        env.use_cwnd(True)
        check_episode(env, d, i, fixed=fixed)
        assert env.rng_draws == int(d["rng"][i][1])
        env.close()
    if name == "cwnd_pm1":
        assert (d["steps"][..., 0] == 0).any()      # some MIs are fully blocked by the window ...
    else:
        assert d["cwnd"].max() == 5000              # ... and the window can open up to MAX_CWND


@pytest.mark.parametrize("name", ["noise_pm1", "noise_fixed_q1", "noise_fixed_lossy", "noise_fixed_deepq"])
def test_use_latency_noise_engine_option_bit_exact(name):
    """The reference's dormant USE_LATENCY_NOISE option (ns:51-52, 150-151, 171-172): every link latency is
    multiplied by random.uniform(1.0, 1.1) -- one more draw of the shared stream per hop, taken before the
    loss decision -- so packets overtake each other on both hops."""
    d = load(name)
    feats = [str(f) for f in d["features"]]
    fixed = tuple(d["fixed"]) if "fixed" in d else None
    for i in range(d["seed"].shape[0]):
        env = run_case_mt(d, i, int(d["history_len"]), feats)
        env.use_latency_noise(True, 1.1)
        check_episode(env, d, i, fixed=fixed)
        assert env.rng_draws == int(d["rng"][i][1])
        env.close()
    # the fixtures are not the noiseless engine's numbers with other seeds: three draws per packet
    plain = load("fixed_q1" if "fixed" in d else "default_pm1")
    per_packet = (d["rng"][:, 1] - d["rng"][:, 0]) / d["steps"][:, :, 0].sum(axis=1)
    per_packet_plain = (plain["rng"][:, 1] - plain["rng"][:, 0]) / plain["steps"][:, :, 0].sum(axis=1)
    assert (per_packet > 2.5).all() and (per_packet_plain < 1.5).all()


def test_two_sender_engine_bit_exact():
    d = load("two_sender")
    for i in range(d["seed"].shape[0]):
        bw, lat, queue, loss, r0, r1, run_dur0 = d["params"][i]
        env = oracle.OracleEnv(2, 10, oracle.DEFAULT_FEATURES)
        env.rng_mt(int(d["seed"][i]), skip=6)
        env.set_params(bw, lat, queue, loss, [r0, r1])
        env.reset()
        assert env.cur_time == d["warm"][i][0] and env.heap_len == int(d["warm"][i][1])
        for t in range(d["actions"].shape[1]):
            obs, rew, done, _ = env.step(d["actions"][i, t])
            for s in range(2):
                assert np.array_equal(env.last_row[s], d["steps"][i, s, t]), (i, s, t)
                assert np.array_equal(obs[s][-3:], d["obs_tail"][i, s, t])
        assert env.rng_draws == int(d["rng"][i][1])
        env.close()


def test_two_sender_engine_with_another_observation_shape_bit_exact():
    """tests/golden/two_sender_allfeat_h3.npz: two senders, all 12 features, three intervals of history -- every step's whole
    observation of both senders (the history / observation writer with S = 2, F != 3, H != 10)."""
    d = load("two_sender_allfeat_h3")
    feats, H = [str(f) for f in d["features"]], int(d["history_len"])
    assert len(feats) == 12 and H == 3
    for i in range(d["seed"].shape[0]):
        bw, lat, queue, loss, r0, r1, run_dur0 = d["params"][i]
        env = oracle.OracleEnv(2, H, feats)
        env.rng_mt(int(d["seed"][i]), skip=6)
        env.set_params(bw, lat, queue, loss, [r0, r1])
        env.reset()
        assert env.cur_time == d["warm"][i][0] and env.heap_len == int(d["warm"][i][1])
        for t in range(d["actions"].shape[1]):
            obs, rew, done, _ = env.step(d["actions"][i, t])
            for s in range(2):
                assert np.array_equal(env.last_row[s], d["steps"][i, s, t]), (i, s, t)
                assert np.array_equal(obs[s], d["obs_full"][i, s, t]), (i, s, t)
        assert env.rng_draws == int(d["rng"][i][1])
        env.close()


def test_survey_kats_seed0():
    """The numbers SURVEY.md section 8(c) quotes for seed 0."""
    d = load("default_pm1")
    env = run_case_mt(d, 0, 10, oracle.DEFAULT_FEATURES)
    env.reset()
    p = env.params()
    assert p[0] == 261.97365498016575 and p[1] == 0.4027093650656477 and round(p[2]) == 12
    assert p[3] == 0.023829847707617792 and p[4] == 261.98896664503104 and p[5] == 1.208128095196943
    assert env.cur_time == 2.419949237247849 and env.heap_len == 213
    acts = np.random.RandomState(0).uniform(-1, 1, 400)
    assert acts[0] == 0.0976270078546495
    total = 0.0
    for t in range(400):
        obs, rew, done, _ = env.step([acts[t]])
        total += rew
        if t == 0:
            r = env.last_row[0]
            assert r[3] == 262.62839661764315 and tuple(r[:3]) == (317, 310, 8)
            assert obs[-3:].tolist() == [0.00010344666874238566, 1.0, 1.0258899676375404]
            assert rew == 1.69943973870502 and r[5] == 0.40274471221925007
    assert done and env.cur_time == 167.20676843687704
    assert total == 625.6196947931776


def test_step_before_reset_is_an_error():
    env = oracle.OracleEnv()
    with pytest.raises(TypeError):
        env.step([0.0])


@pytest.mark.parametrize("name", ["cwnd_noise_pm1", "cwnd_noise_grow"])
def test_both_engine_options_together_bit_exact(name):
    """USE_CWND and USE_LATENCY_NOISE are module globals of the reference engine (ns:51-54): switched on together they
    apply together -- a SEND the window blocks still takes its latency-noise draw and its loss draw (ns:158-175)."""
    d = load(name)
    feats = [str(f) for f in d["features"]]
    for i in range(d["seed"].shape[0]):
        env = run_case_mt(d, i, int(d["history_len"]), feats)
        env.use_cwnd(True)
        env.use_latency_noise(True, 1.1)
        check_episode(env, d, i)
        assert env.rng_draws == int(d["rng"][i][1])
        env.close()


@pytest.mark.parametrize("name,cwnd,noise", [("two_sender_cwnd", True, False), ("two_sender_noise", False, True),
                                             ("two_sender_cwnd_noise", True, True)])
def test_engine_options_with_two_senders_bit_exact(name, cwnd, noise):
    """The dormant options are module globals the engine reads for whatever senders it holds (ns:51-54, 150-175): two
    senders on the bottleneck, every sender with its own window and its own [rate action, cwnd action]."""
    d = load(name)
    for i in range(d["seed"].shape[0]):
        bw, lat, queue, loss, r0, r1, run_dur0 = d["params"][i]
        env = oracle.OracleEnv(2, 10, oracle.DEFAULT_FEATURES)
        env.rng_mt(int(d["seed"][i]), skip=6)
        env.set_params(bw, lat, queue, loss, [r0, r1])
        if cwnd:
            env.use_cwnd(True)
        if noise:
            env.use_latency_noise(True, 1.1)
        env.reset()
        assert env.cur_time == d["warm"][i][0] and env.heap_len == int(d["warm"][i][1])
        for t in range(d["actions"].shape[1]):
            a = np.stack([d["actions"][i, t], d["cwnd_actions"][i, t]], axis=1) if cwnd else d["actions"][i, t]
            obs, rew, done, _ = env.step(a)
            for s in range(2):
                assert np.array_equal(env.last_row[s], d["steps"][i, s, t]), (i, s, t)
                assert np.array_equal(obs[s][-3:], d["obs_tail"][i, s, t])
                if cwnd:
                    assert env.cwnd(s) == int(d["cwnd"][i, t, s])
        assert env.rng_draws == int(d["rng"][i][1])
        env.close()
    if cwnd:   # the windows did limit the senders: fewer packets than the same links carry without the option
        assert (d["cwnd"] != 25).any()
