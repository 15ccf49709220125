"""The heap-free restatement of USE_LATENCY_NOISE (tests/models/noise_sorting_model.py, the numpy design study of pcc_noise_sorted.hip: counts, two sorts and a scan per interval)
against the oracle's event loop: every observation, reward, count and clock bit for bit.  CPU only."""
import numpy as np
import pytest

from oracle.pcc_oracle_py import PyOracleEnv
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "models"))
from noise_sorting_model import SortedNoiseEnv   # noqa: E402


def run_pair(seed, n_steps, scale=1.0, fixed=None):
    a = PyOracleEnv(seed=seed, latency_noise=1.1, fixed=fixed)
    b = SortedNoiseEnv(seed=seed, latency_noise=1.1, fixed=fixed)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    acts = np.random.RandomState(seed).uniform(-1, 1, n_steps) * scale
    for t in range(n_steps):
        oa, ra, da, _ = a.step(acts[t])
        ob, rb, db, _ = b.step(acts[t])
        assert (a.sent, a.acked, a.lost) == (b.sent, b.acked, b.lost), (seed, t)
        assert a.now == b.now and a.q == b.q and a.tq == b.tq, (seed, t)
        assert a.rtts[0] == b.rtts[0], (seed, t)
        assert np.array_equal(oa, ob) and ra == rb and da == db, (seed, t)
        assert a.draws == b.draws, (seed, t)
    return b


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 7])
def test_random_links_match_the_event_loop(seed):
    run_pair(seed, 120)


def test_rates_pushed_up_and_down_match_and_the_blocks_are_wide():
    b = run_pair(11, 150, scale=8.0)                     # large actions: rates from the floor to the ceiling
    b = run_pair(5, 100, fixed=(400.0, 0.3, 2000, 0.02, 900.0))   # a fast sender on a long link: many packets in flight
    assert max(b.block_sizes) >= 100                      # hundreds of SENDs with no dependence on each other


def test_lossy_shallow_queue_matches():
    run_pair(3, 100, fixed=(150.0, 0.08, 2, 0.04, 200.0))


# ---- one or two senders on the link (tests/models/noise_sorting2_model.py)
from noise_sorting2_model import SortedNoiseEnvS   # noqa: E402


def run_pair_s(seed, n_steps, n_senders, scale=1.0, fixed=None):
    a = PyOracleEnv(seed=seed, latency_noise=1.1, fixed=fixed, n_senders=n_senders)
    b = SortedNoiseEnvS(seed=seed, latency_noise=1.1, fixed=fixed, n_senders=n_senders)
    oa, ob = a.reset(), b.reset()
    assert np.array_equal(oa, ob)
    acts = np.random.RandomState(seed).uniform(-1, 1, (n_steps, n_senders)) * scale
    for t in range(n_steps):
        act = acts[t] if n_senders > 1 else acts[t, 0]
        oa, ra, da, _ = a.step(act)
        ob, rb, db, _ = b.step(act)
        assert (a.sent, a.acked, a.lost) == (b.sent, b.acked, b.lost), (seed, t)
        assert a.now == b.now and a.q == b.q and a.tq == b.tq, (seed, t)
        assert a.rtts == b.rtts, (seed, t)
        assert np.array_equal(oa, ob) and ra == rb and da == db, (seed, t)
        assert a.draws == b.draws, (seed, t)
    return b


@pytest.mark.parametrize("seed", [0, 1, 4])
def test_two_senders_match_the_event_loop(seed):
    run_pair_s(seed, 100, 2)


def test_two_senders_large_actions_and_a_fast_pair_on_a_long_link():
    run_pair_s(9, 120, 2, scale=8.0)
    b = run_pair_s(6, 80, 2, fixed=(450.0, 0.3, 3000, 0.01, 700.0, 300.0))
    assert max(b.block_sizes) >= 100


def test_the_general_restatement_with_one_sender():
    run_pair_s(2, 100, 1)
