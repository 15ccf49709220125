"""ctypes binding of the C ABI declared in include/pcc_sim.h.

There is no CPU fallback: if the HIP library is missing or cannot be loaded, importing the
binding's entry points raises, loudly, with the build command to run."""
import ctypes
import os

from .build import build_library, library_path

PCC_RNG_PHILOX, PCC_RNG_TRACE = 0, 1
PCC_FLAG_RING_OVERFLOW, PCC_FLAG_TRACE_OVERRUN, PCC_FLAG_INTERNAL, PCC_FLAG_POOL_EXHAUSTED, PCC_FLAG_BAD_PARAMS = 1, 2, 4, 8, 16
PCC_FLAG_BAD_ACTION, PCC_FLAG_TIME_RANGE = 32, 64
PCC_STEP_COLS = 19
STEP_COLUMNS = ["sent", "acked", "lost", "rate", "cur_time", "run_dur", "reward",
                "send rate", "recv rate", "recv dur", "send dur", "avg latency", "loss ratio",
                "ack latency inflation", "sent latency inflation", "conn min latency",
                "latency increase", "latency ratio", "send ratio"]

# name -> (field id, torch dtype name, per-sender?)
FIELDS = {
    "bw": (0, "float64", False), "dl": (1, "float64", False), "lr": (2, "float64", False),
    "maxq": (3, "float64", False), "queue_delay": (4, "float64", False), "queue_time": (5, "float64", False),
    "now": (6, "float64", False), "run_dur": (7, "float64", False), "steps": (8, "int32", False),
    "episode": (9, "int32", False), "flags": (10, "int32", False), "rate": (11, "float64", True),
    "rate0": (12, "float64", True), "next_send": (13, "float64", True), "min_lat": (14, "float64", True),
    "acc_head": (15, "int32", True), "acc_tail": (16, "int32", True), "drop_head": (17, "int32", True),
    "drop_tail": (18, "int32", True), "ep_return": (19, "float64", True), "last_return": (20, "float64", True),
    "total_sent": (21, "int64", False), "ring_tier": (22, "uint8", True), "cwnd": (23, "int32", True),
}

# every symbol include/pcc_sim.h declares
SYMBOLS = ["pcc_last_error", "pcc_create", "pcc_destroy", "pcc_set_link_params", "pcc_set_param_ranges",
           "pcc_set_rng", "pcc_set_seed", "pcc_set_tuning", "pcc_set_ring_pools", "pcc_set_cwnd_mode", "pcc_set_latency_noise", "pcc_set_delta_scale", "pcc_set_max_steps", "pcc_reset", "pcc_step", "pcc_step_many", "pcc_step_send",
           "pcc_step_retire",
           "pcc_get_state", "pcc_restart_stats", "pcc_fused_steps", "pcc_debug_addresses", "pcc_metric_info", "pcc_device_bytes", "pcc_debug_timeline", "pcc_debug_pass_stats",
           "pcc_policy_act", "pcc_ppo_scratch_floats", "pcc_ppo_minibatch_step", "pcc_gae"]


class PccError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("pcc_sim error %d: %s" % (code, message))
        self.code = code


_lib = None


def lib():
    """The loaded HIP library (loads it on first use; never falls back to anything else)."""
    global _lib
    if _lib is not None:
        return _lib
    # tools that ask for the profiling hooks (PCC_DEBUG_TIMELINE / PCC_DEBUG_SKIP in the environment) get the -DPCC_PROFILE=1
    # build; the product library has none of them compiled in and reads neither variable
    profile = bool(os.environ.get("PCC_DEBUG_TIMELINE") or os.environ.get("PCC_DEBUG_SKIP"))
    path = build_library(profile=True) if profile else library_path()
    if os.environ.get("PCC_SIM_LIBRARY"):   # tools/ experiments: another build of the same sources (A/B runs of compile-time variants)
        path = os.environ["PCC_SIM_LIBRARY"]
    if not os.path.exists(path):
        raise ImportError(
            "%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950).  This package has no CPU fallback." % path)
    L = ctypes.CDLL(path)
    vp, i32, i64, u32, u64, dbl = (ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_uint32,
                                   ctypes.c_uint64, ctypes.c_double)
    L.pcc_last_error.restype = ctypes.c_char_p
    L.pcc_last_error.argtypes = []
    L.pcc_create.argtypes = [i64, i32, i32, ctypes.POINTER(ctypes.c_int32), i32, u64, u32, u32, i32,
                             ctypes.POINTER(vp)]
    L.pcc_destroy.restype = None
    L.pcc_destroy.argtypes = [vp]
    L.pcc_set_link_params.argtypes = [vp, vp, vp, vp, vp, vp]
    L.pcc_set_param_ranges.argtypes = [vp, ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    L.pcc_set_rng.argtypes = [vp, i32, vp, i64]
    L.pcc_set_seed.argtypes = [vp, u64]
    L.pcc_set_tuning.argtypes = [vp, i32, dbl]
    L.pcc_set_tuning.restype = i32
    L.pcc_set_ring_pools.argtypes = [vp, u32, u32, u32]
    L.pcc_set_ring_pools.restype = i32
    L.pcc_set_cwnd_mode.argtypes = [vp, i32]
    L.pcc_set_cwnd_mode.restype = i32
    L.pcc_set_latency_noise.argtypes = [vp, i32, dbl]
    L.pcc_set_latency_noise.restype = i32
    L.pcc_set_delta_scale.argtypes = [vp, dbl]
    L.pcc_set_max_steps.argtypes = [vp, i32]
    L.pcc_reset.argtypes = [vp, vp, vp, vp]
    L.pcc_step.argtypes = [vp, vp, i32, vp, vp, vp, vp, i32, vp]
    L.pcc_step_many.argtypes = [vp, vp, i32, i32, vp, vp, vp, vp, i32, vp]
    L.pcc_step_send.argtypes = [vp, vp, i32, vp]
    L.pcc_step_retire.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    L.pcc_get_state.argtypes = [vp, i32, vp, vp]
    L.pcc_restart_stats.restype = i32
    L.pcc_restart_stats.argtypes = [vp, vp, vp]
    L.pcc_fused_steps.restype = i32
    L.pcc_fused_steps.argtypes = [vp, vp]
    L.pcc_debug_addresses.restype = i32
    L.pcc_debug_addresses.argtypes = [vp, vp]
    L.pcc_metric_info.argtypes = [i32, ctypes.POINTER(dbl), ctypes.POINTER(dbl), ctypes.POINTER(dbl)]
    L.pcc_device_bytes.restype = i64
    L.pcc_device_bytes.argtypes = [vp]
    L.pcc_debug_timeline.restype = i64
    L.pcc_debug_timeline.argtypes = [vp, vp, i64]
    L.pcc_policy_act.restype = i32
    L.pcc_policy_act.argtypes = [vp, i64, i32, vp, i32, i32, vp, vp, vp, vp, vp, vp]
    f32 = ctypes.c_float
    L.pcc_ppo_scratch_floats.restype = i32
    L.pcc_ppo_scratch_floats.argtypes = [i32, i32, i32]
    L.pcc_ppo_minibatch_step.restype = i32
    L.pcc_ppo_minibatch_step.argtypes = [vp, vp, vp, vp, vp, vp, i64, i64, i32, i32, i32, vp, vp, vp, i32, f32, f32, f32, f32,
                                         f32, f32, vp, vp, vp, vp]
    L.pcc_gae.restype = i32
    L.pcc_gae.argtypes = [vp, vp, vp, vp, i32, i64, f32, f32, vp, vp, vp]
    L.pcc_debug_pass_stats.restype = i32
    L.pcc_debug_pass_stats.argtypes = [vp, vp, i32]
    for fn in ("pcc_create", "pcc_set_link_params", "pcc_set_param_ranges", "pcc_set_rng", "pcc_set_seed",
               "pcc_set_delta_scale", "pcc_set_max_steps", "pcc_reset", "pcc_step", "pcc_step_many", "pcc_step_send",
               "pcc_step_retire", "pcc_get_state", "pcc_metric_info"):
        getattr(L, fn).restype = i32
    _lib = L
    return L


def check(code):
    if code != 0:
        raise PccError(code, lib().pcc_last_error().decode("utf-8", "replace"))
