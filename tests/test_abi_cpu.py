"""CPU-side checks of the product package: the C-ABI library loads and exports every symbol
include/pcc_sim.h declares, the host-side mirror of the reference interface behaves, and the
product refuses to run without a GPU instead of falling back to anything."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import pcc_rl_amd
from pcc_rl_amd import native

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libpath():
    return pcc_rl_amd.build_library()


def declared_functions():
    names = set()
    for header in ("pcc_sim.h", "pcc_policy.h"):     # every header under include/
        text = open(os.path.join(ROOT, "include", header)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        names |= set(re.findall(r"\b(pcc_[a-z_]+)\s*\(", text))
    return sorted(names)


def test_header_and_binding_agree():
    assert declared_functions() == sorted(native.SYMBOLS)


def test_library_exports_every_declared_symbol(libpath):
    L = ctypes.CDLL(libpath)
    for name in declared_functions():
        assert hasattr(L, name), name


def test_metric_info_matches_python_table(libpath):
    L = native.lib()
    for i, name in enumerate(pcc_rl_amd.METRIC_NAMES):
        mn, mx, sc = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        assert L.pcc_metric_info(i, ctypes.byref(mn), ctypes.byref(mx), ctypes.byref(sc)) == 0
        assert (mn.value, mx.value, sc.value) == pcc_rl_amd.metric_info(name)
    assert L.pcc_metric_info(99, None, None, None) < 0
    assert b"out of range" in L.pcc_last_error()


def test_observation_bounds_like_reference():
    # ns:382-388: Box(tile([-1, 1, 0], 10), tile([10, 1e4, 1e3], 10), float32)
    f = pcc_rl_amd.DEFAULT_FEATURES
    assert pcc_rl_amd.get_min_obs_vector(f).tolist() == [-1.0, 1.0, 0.0]
    assert pcc_rl_amd.get_max_obs_vector(f).tolist() == [10.0, 10000.0, 1000.0]
    assert pcc_rl_amd.feature_ids(f) == [7, 10, 11]
    with pytest.raises(KeyError):
        pcc_rl_amd.feature_ids("no such metric")


def test_arg_or_default_casts_like_reference():
    from pcc_rl_amd import config
    table = config._scan(["prog", "--history-len=5", "--delta-scale=0.05", "--flag", "--name=x"])
    old, config._ARGS = config._ARGS, table
    try:
        assert config.arg_or_default("--history-len", default=10) == 5
        assert config.arg_or_default("--delta-scale", 0.025) == 0.05
        assert config.arg_or_default("--name", "d") == "x"
        assert config.arg_or_default("--flag") is True
        assert config.arg_or_default("--absent", 3) == 3
    finally:
        config._ARGS = old


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_silent_cpu_fallback(libpath):
    with pytest.raises((RuntimeError, ValueError)):
        pcc_rl_amd.BatchedNetworkEnv(4, device="cuda")
    with pytest.raises(ValueError):
        pcc_rl_amd.BatchedNetworkEnv(4, device="cpu")
    # straight through the C ABI: create must fail with ENODEV, not succeed on some host path
    L = native.lib()
    fids = (ctypes.c_int32 * 3)(7, 10, 11)
    h = ctypes.c_void_p()
    rc = L.pcc_create(4, 1, 10, fids, 3, 0, 0, 0, -1, ctypes.byref(h))
    assert rc == -2 and not h.value
    assert b"no CPU path" in L.pcc_last_error() or b"gfx950" in L.pcc_last_error()


def test_create_rejects_bad_arguments(libpath):
    L = native.lib()
    fids = (ctypes.c_int32 * 3)(7, 10, 11)
    h = ctypes.c_void_p()
    assert L.pcc_create(0, 1, 10, fids, 3, 0, 0, 0, -1, ctypes.byref(h)) == -1
    assert L.pcc_create(4, 3, 10, fids, 3, 0, 0, 0, -1, ctypes.byref(h)) == -1
    assert L.pcc_create(4, 1, 10, fids, 3, 0, 0, 1000, -1, ctypes.byref(h)) == -1   # not a power of two
    bad = (ctypes.c_int32 * 1)(12)
    assert L.pcc_create(4, 1, 10, bad, 1, 0, 0, 0, -1, ctypes.byref(h)) == -1


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "pcc-rl_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, fn)).read()
                assert "import oracle" not in text and "libpcc_oracle" not in text, fn


def test_hot_kernels_do_not_spill():
    """The compiler's resource report of the library as built (pcc-rl_amd/build.py keeps it next to the .so): the kernels
    of BASELINE config 3's step -- and the restart and small-batch kernels -- use no scratch memory and spill no vector
    register.  Round 3's send kernel spilled 56-160 bytes per lane and its results depended on how the spill code came
    out (two builds that only added exact code broke paths they did not touch); since round 4 no lane state is kept
    across the wave passes, and this test keeps it that way."""
    import json
    import os
    from pcc_rl_amd import build as pbuild
    path = pbuild.library_path() + ".resources.json"
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(pbuild.library_path()):
        pbuild.build_library(force=True)
    with open(path) as f:
        res = json.load(f)
    for name in ("send_kernel<1, false>", "send_kernel<1, true>", "retire_kernel<1, false>", "send_restart_kernel<1, false>",
                 "send_restart_kernel<2, false>", "step_small_kernel<1, false>"):
        assert name in res, sorted(res)
        r = res[name]
        assert r["vgpr_spills"] == 0, (name, r)
        if name != "send_kernel<1, true>":   # (the trace build reserves 20 bytes it never touches: no scratch instruction in its code)
            assert r["scratch"] == 0, (name, r)
    assert res["send_kernel<1, false>"]["occupancy"] == 4 and res["retire_kernel<1, false>"]["occupancy"] == 4
    # the one-launch step (experimental, off by default): the register budget and the occupancy of the two launches; what it spills
    # (one or two dozen registers around its retire loop, none in a hot loop: pcc-rl_amd/csrc/pcc_fused.hip) stays small
    f = res["step_fused_kernel<1, false>"]
    assert f["vgprs"] <= 128 and f["occupancy"] == 4 and f["scratch"] <= 96 and f["vgpr_spills"] <= 32, f
    # latency noise without the event loop: nothing in scratch; the arrays of the first instance leave room for a dozen workgroups
    # per compute unit, those of the second for three (160 KB of LDS per compute unit)
    small, large = res["noise_sorted_kernel<256, 128, 128, false, 64>"], res["noise_sorted_kernel<1024, 512, 256, true, 256>"]
    for r in (small, large):
        assert r["scratch"] == 0 and r["vgpr_spills"] == 0, r
    assert small["lds"] <= 13 * 1024 and large["lds"] <= 48 * 1024, (small, large)


def test_no_built_binary_is_tracked():
    """History holds sources only: shared libraries, objects and executables are built on the spot (they still travel to
    the GPU box with the snapshot).  Skipped outside a git checkout."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    try:
        files = subprocess.run(["git", "ls-files"], cwd=root, capture_output=True, text=True, check=True).stdout.split("\n")
    except Exception:
        pytest.skip("not a git checkout")
    elf = []
    for f in files:
        p = os.path.join(root, f)
        if f and os.path.isfile(p):
            with open(p, "rb") as fh:
                if fh.read(4) == b"\x7fELF":
                    elf.append(f)
    assert not elf, "built binaries in the index: %s" % elf
