"""The closed-form wave pass of the send half (pcc_sim.hip: heavy_mi), as a CPU model with the kernel's own
integer / floating-point operations, against the plain per-packet recurrence (ns:66-84): fuzzed link states
and the interval-start states of real episodes.  Any mismatch is a counter-example for the kernel's
preconditions, found without a GPU."""
import ctypes
import os
import subprocess

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "models", "send_pass_model.c")
LIB = os.path.join(HERE, "models", "libsend_pass_model.so")


@pytest.fixture(scope="module")
def model():
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fopenmp", SRC, "-o", LIB, "-lm"])
    L = ctypes.CDLL(LIB)
    L.pcc_model_fuzz.restype = ctypes.c_long
    L.pcc_model_fuzz.argtypes = [ctypes.c_long, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    L.pcc_model_episodes.restype = ctypes.c_long
    L.pcc_model_set_lanes.argtypes = [ctypes.c_int]
    L.pcc_model_fuzz_straddle.argtypes = [ctypes.c_int]
    return L


# lanes of one pass: 64 = one wavefront (heavy_mi<.., 1>), 256 = a team of four wavefronts (heavy_mi<.., 4>)
LANES = [64, 256]


@pytest.mark.parametrize("lanes", LANES)
def test_fuzzed_link_states(model, lanes):
    assert model.pcc_model_set_lanes(lanes) == 0
    stats = (ctypes.c_uint64 * 16)()
    bad = model.pcc_model_fuzz(20000, 12345 + lanes, stats)
    assert bad == 0
    s = np.array(list(stats), dtype=np.uint64)
    assert s[:3].sum() > 0          # the closed-form regimes were exercised, not only the serial pass


@pytest.mark.parametrize("lanes", LANES)
def test_interval_starts_of_real_episodes(model, lanes):
    assert model.pcc_model_set_lanes(lanes) == 0
    model.pcc_model_episodes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64)]
    stats, hist = (ctypes.c_uint64 * 16)(), (ctypes.c_uint64 * 32)()
    bad = model.pcc_model_episodes(48, 120, 7, 0, 64, stats, hist)
    assert bad == 0
    assert sum(hist) > 0


def test_fuzzed_queue_limits_just_above_a_power_of_two(model):
    """Regime C (heavy_mi: the full queue straddles a power of two, two rounding grids): half of the fuzzed links get a
    queue limit just above 0.25 .. 16 s and an overdriving sender; the passes must still equal the plain recurrence, and
    the regime must carry packets."""
    assert model.pcc_model_set_lanes(64) == 0
    model.pcc_model_fuzz_straddle(1)
    try:
        stats = (ctypes.c_uint64 * 16)()
        bad = model.pcc_model_fuzz(20000, 4242, stats)
    finally:
        model.pcc_model_fuzz_straddle(0)
    assert bad == 0
    assert stats[14] > 1000 and stats[15] > 100 * stats[14]     # regime-C passes, well filled


SRC2 = os.path.join(HERE, "models", "send_pass2_model.c")
LIB2 = os.path.join(HERE, "models", "libsend_pass2_model.so")


@pytest.fixture(scope="module")
def model2():
    if not os.path.exists(LIB2) or os.path.getmtime(LIB2) < os.path.getmtime(SRC2):
        subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", SRC2, "-o", LIB2, "-lm"])
    L = ctypes.CDLL(LIB2)
    L.pcc_model2_fuzz.restype = ctypes.c_long
    L.pcc_model2_fuzz.argtypes = [ctypes.c_long, ctypes.c_uint64, ctypes.POINTER(ctypes.c_uint64)]
    L.pcc_model2_set_shape.argtypes = [ctypes.c_int, ctypes.c_int]
    return L


@pytest.mark.parametrize("lanes,per_lane", [(64, 1), (64, 4)])
def test_two_sender_token_pass_model(model2, lanes, per_lane):
    """The token pass of the two-sender wave path (heavy_mi2: a backlogged queue in one binade, accept decisions by a
    token bucket with uneven arrivals over the merged stream) against the plain merged recurrence on fuzzed link states:
    pairs that overdrive the link, queue limits around powers of two, young and old clocks, equal send times.  64 x 1
    is the kernel's pass; 64 x 4 the variant with one Philox block per lane."""
    assert model2.pcc_model2_set_shape(lanes, per_lane) == 0
    stats = (ctypes.c_uint64 * 4)()
    bad = model2.pcc_model2_fuzz(6000, 777 + per_lane, stats)
    assert bad == 0
    passes, committed, plain = int(stats[0]), int(stats[1]), int(stats[2])
    assert passes > 1000 and committed > 5 * plain // 10      # the pass carried a good part of the packets
