"""Builds pcc-rl_amd/lib/libpcc_sim.so from csrc/*.hip with hipcc for gfx950.
hipcc cross-compiles without a GPU, so this also runs in the GPU-less build container."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
# one translation unit per kernel family (each kernel is register-allocated on its own) + the C ABI + the PPO caller's kernels
UNITS = ["pcc_sim.hip", "pcc_send.hip", "pcc_send_restart.hip", "pcc_retire.hip", "pcc_small.hip", "pcc_fused.hip", "pcc_noise_sorted.hip",
         "pcc_policy.hip", "pcc_ppo.hip"]
SRCS = [os.path.join(CSRC, u) for u in UNITS]
HEADERS = [os.path.join(CSRC, h) for h in ("pcc_dev.h", "pcc_kernels.h", "pcc_wave_pass.h", "pcc_send_item.h", "pcc_send_bodies.h", "pcc_retire_env.h")]
INCLUDE = os.path.join(ROOT, "include")
LIB_DIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIB_DIR, "libpcc_sim.so")
# the same sources with -DPCC_PROFILE=1: per-item timelines, pass counters and the "skip" switches of tools/ -- never the
# product (those switches drop work: wrong results, timing only)
PROFILE_LIB = os.path.join(LIB_DIR, "libpcc_sim_prof.so")

HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC"]


def library_path(profile=False):
    return PROFILE_LIB if profile else LIB


def _stale(lib):
    if not os.path.exists(lib):
        return True
    newest = max(os.path.getmtime(p) for p in SRCS + HEADERS + [os.path.join(INCLUDE, "pcc_sim.h"), os.path.join(INCLUDE, "pcc_policy.h")])
    return os.path.getmtime(lib) < newest


# variant builds the GPU tests run the parity suite through (tests/test_variants.py): code paths that are exact by
# construction but once broke in code generation: the adaptive trigger of regime C; the restart kernel and the
# small-batch kernel cut for 4 wavefronts per SIMD (128 registers: they spill)
VARIANTS = {
    "adaptc": ["-DPCC_ADAPTIVE_C=1"],
    "tight": ["-DPCC_RESTART_OCC=4", "-DPCC_SMALL_OCC=4"],
}


def variant_path(name):
    return os.path.join(LIB_DIR, "libpcc_sim_var_%s.so" % name)


def _parse_resources(text):
    """kernel -> {vgprs, sgprs, scratch, occupancy, vgpr_spills, sgpr_spills} from -Rpass-analysis=kernel-resource-usage."""
    import re
    out, cur = {}, None
    keys = {"VGPRs:": "vgprs", "TotalSGPRs:": "sgprs", "ScratchSize [bytes/lane]:": "scratch", "Occupancy [waves/SIMD]:": "occupancy",
            "VGPRs Spill:": "vgpr_spills", "SGPRs Spill:": "sgpr_spills", "LDS Size [bytes/block]:": "lds"}
    for line in text.splitlines():
        m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?)\s+\[-Rpass-analysis", line) or re.search(r"remark:\s+(.*?)\s+\[-Rpass-analysis", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()
            out[cur] = {}
        elif cur is not None:
            for k, name in keys.items():
                if t.startswith(k):
                    try:
                        out[cur][name] = int(t[len(k):].strip())
                    except ValueError:
                        pass
    return out


def build_library(force=False, verbose=False, profile=False, extra_flags=(), out=None):
    """Compile the HIP library if it is missing or older than its sources; returns its path.
    profile=True builds the tools' variant (libpcc_sim_prof.so, -DPCC_PROFILE=1) instead.  The translation units are
    compiled side by side (one hipcc process each) and linked into one shared object; extra_flags / out are for variant
    builds (e.g. -DPCC_ADAPTIVE_C=1 into another file).  The compiler's per-kernel resource report (registers, scratch,
    spills) is kept next to the library as <lib>.resources.json: tests/test_abi_cpu.py reads it."""
    import json
    lib = out or library_path(profile)
    if not force and not _stale(lib):
        return lib
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        raise RuntimeError("hipcc not found; cannot build %s" % lib)
    os.makedirs(LIB_DIR, exist_ok=True)
    obj_dir = os.path.join(LIB_DIR, "obj_" + os.path.splitext(os.path.basename(lib))[0])
    os.makedirs(obj_dir, exist_ok=True)
    flags = (HIPCC_FLAGS + (["-DPCC_PROFILE=1"] if profile else []) + list(extra_flags) +
             ["-I", INCLUDE, "-I", CSRC, "-Rpass-analysis=kernel-resource-usage"])
    procs, objs = [], []
    for src in SRCS:
        obj = os.path.join(obj_dir, os.path.splitext(os.path.basename(src))[0] + ".o")
        objs.append(obj)
        cmd = [hipcc] + flags + ["-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, universal_newlines=True)))
    resources = {}
    for cmd, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("%s failed:\n%s" % (" ".join(cmd), err[-4000:]))
        resources.update(_parse_resources(err))
    link = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", lib]
    if verbose:
        print(" ".join(link))
    subprocess.check_call(link)
    try:
        names = subprocess.run(["c++filt"], input="\n".join(resources), stdout=subprocess.PIPE, universal_newlines=True).stdout.split("\n")
        resources = {n.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""): v for n, v in zip(names, resources.values())}
    except OSError:
        pass
    with open(lib + ".resources.json", "w") as f:
        json.dump(resources, f, indent=1, sort_keys=True)
    with open(lib + ".build_info.json", "w") as f:
        json.dump(_source_stamp(extra_flags, profile), f, indent=1, sort_keys=True)
    return lib


def _source_stamp(extra_flags=(), profile=False):
    """What a built library was compiled from: the commit of the tree (and whether csrc/ or include/ differed from it) -- the GPU
    box gets the built .so but no .git, and every measurement file names the code it measured -- plus a hash of the sources."""
    import hashlib
    import time
    h = hashlib.sha256()
    for p in sorted(SRCS + HEADERS + [os.path.join(INCLUDE, "pcc_sim.h"), os.path.join(INCLUDE, "pcc_policy.h")]):
        with open(p, "rb") as f:
            h.update(f.read())
    stamp = {"sources_sha16": h.hexdigest()[:16], "flags": HIPCC_FLAGS + (["-DPCC_PROFILE=1"] if profile else []) + list(extra_flags),
             "built_at": time.strftime("%Y-%m-%dT%H:%M:%SZ", time.gmtime()), "commit": None, "dirty": None}
    try:
        stamp["commit"] = subprocess.run(["git", "rev-parse", "--short=12", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                         universal_newlines=True, timeout=10).stdout.strip() or None
        if stamp["commit"]:
            stamp["dirty"] = bool(subprocess.run(["git", "status", "--porcelain", "--", "pcc-rl_amd/csrc", "include"], cwd=ROOT,
                                                 stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, universal_newlines=True, timeout=10).stdout.strip())
    except (OSError, subprocess.SubprocessError):
        pass
    return stamp


def build_info(lib=None):
    """The stamp build_library() left next to `lib` (default: the product library); {} if there is none."""
    import json
    try:
        with open((lib or LIB) + ".build_info.json") as f:
            return json.load(f)
    except (OSError, ValueError):
        return {}


def build_variants(force=False, verbose=False):
    """The variant libraries of VARIANTS (see there); returns {name: path}."""
    return {name: build_library(force=force, verbose=verbose, extra_flags=flags, out=variant_path(name)) for name, flags in VARIANTS.items()}
