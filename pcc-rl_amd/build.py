"""Builds pcc-rl_amd/lib/libpcc_sim.so from csrc/pcc_sim.hip with hipcc for gfx950.
hipcc cross-compiles without a GPU, so this also runs in the GPU-less build container."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "pcc_sim.hip")
SRCS = [SRC, os.path.join(HERE, "csrc", "pcc_policy.hip"), os.path.join(HERE, "csrc", "pcc_ppo.hip")]
INCLUDE = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libpcc_sim.so")
# the same sources with -DPCC_PROFILE=1: per-item timelines, pass counters and the "skip" switches of tools/ -- never the
# product (those switches drop work: wrong results, timing only)
PROFILE_LIB = os.path.join(LIB_DIR, "libpcc_sim_prof.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def library_path(profile=False):
    return PROFILE_LIB if profile else LIB


def _stale(lib):
    if not os.path.exists(lib):
        return True
    newest = max(os.path.getmtime(p) for p in SRCS + [os.path.join(INCLUDE, "pcc_sim.h"), os.path.join(INCLUDE, "pcc_policy.h")])
    return os.path.getmtime(lib) < newest


def build_library(force=False, verbose=False, profile=False):
    """Compile the HIP library if it is missing or older than its sources; returns its path.
    profile=True builds the tools' variant (libpcc_sim_prof.so, -DPCC_PROFILE=1) instead."""
    lib = library_path(profile)
    if not force and not _stale(lib):
        return lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build %s" % lib)
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + (["-DPCC_PROFILE=1"] if profile else []) + ["-I", INCLUDE] + SRCS + ["-o", lib]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return lib
