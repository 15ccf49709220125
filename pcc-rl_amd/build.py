"""Builds pcc-rl_amd/lib/libpcc_sim.so from csrc/pcc_sim.hip with hipcc for gfx950.
hipcc cross-compiles without a GPU, so this also runs in the GPU-less build container."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "pcc_sim.hip")
SRCS = [SRC, os.path.join(HERE, "csrc", "pcc_policy.hip")]
INCLUDE = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libpcc_sim.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared"]


def library_path():
    return LIB


def _stale():
    if not os.path.exists(LIB):
        return True
    newest = max(os.path.getmtime(p) for p in SRCS + [os.path.join(INCLUDE, "pcc_sim.h"), os.path.join(INCLUDE, "pcc_policy.h")])
    return os.path.getmtime(LIB) < newest


def build_library(force=False, verbose=False):
    """Compile the HIP library if it is missing or older than its sources; returns its path."""
    if not force and not _stale():
        return LIB
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build %s" % LIB)
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [hipcc] + HIPCC_FLAGS + ["-I", INCLUDE] + SRCS + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB
