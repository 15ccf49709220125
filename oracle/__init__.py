"""ctypes front end of the CPU oracle (oracle/pcc_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT: imported only by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg.  The product
package (pcc-rl_amd/) never imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpcc_oracle.so")

RNG_MT, RNG_TRACE, RNG_PHILOX = 0, 1, 2
MEAN_NUMPY, MEAN_SEQUENTIAL = 0, 1
STEP_COLS = 19
N_METRICS = 12

METRIC_NAMES = ["send rate", "recv rate", "recv dur", "send dur", "avg latency",
                "loss ratio", "ack latency inflation", "sent latency inflation",
                "conn min latency", "latency increase", "latency ratio", "send ratio"]
STEP_COLUMNS = ["sent", "acked", "lost", "rate", "cur_time", "run_dur", "reward"] + METRIC_NAMES
DEFAULT_FEATURES = ("sent latency inflation", "latency ratio", "send ratio")


def feature_ids(names):
    if isinstance(names, str):
        names = names.split(",")
    return [METRIC_NAMES.index(n) for n in names]


def build(force=False):
    """Compile libpcc_oracle.so with gcc if missing or stale."""
    src = os.path.join(HERE, "pcc_oracle.c")
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-B", "-C", HERE, "libpcc_oracle.so"])
    return LIB_PATH


_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)
_u64p = ctypes.POINTER(ctypes.c_uint64)
_u32p = ctypes.POINTER(ctypes.c_uint32)


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(LIB_PATH)
        L.pcc_oracle_create.restype = ctypes.c_void_p
        L.pcc_oracle_create.argtypes = [ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int, ctypes.c_int]
        L.pcc_oracle_destroy.argtypes = [ctypes.c_void_p]
        L.pcc_oracle_rng_mt.argtypes = [ctypes.c_void_p, ctypes.c_uint64]
        L.pcc_oracle_rng_trace.argtypes = [ctypes.c_void_p, _dp, ctypes.c_long]
        L.pcc_oracle_rng_philox.argtypes = [ctypes.c_void_p, ctypes.c_uint64, ctypes.c_uint32]
        L.pcc_oracle_rng_skip.argtypes = [ctypes.c_void_p, ctypes.c_long]
        L.pcc_oracle_rng_draws.restype = ctypes.c_long
        L.pcc_oracle_rng_draws.argtypes = [ctypes.c_void_p]
        L.pcc_oracle_trace_overrun.argtypes = [ctypes.c_void_p]
        L.pcc_oracle_set_params.argtypes = [ctypes.c_void_p] + [ctypes.c_double] * 4 + [_dp]
        L.pcc_oracle_clear_params.argtypes = [ctypes.c_void_p]
        L.pcc_oracle_set_ranges.argtypes = [ctypes.c_void_p, _dp, _dp]
        L.pcc_oracle_reset.argtypes = [ctypes.c_void_p, _dp]
        L.pcc_oracle_step.argtypes = [ctypes.c_void_p, _dp, ctypes.c_double, _dp, _dp, _dp]
        L.pcc_oracle_get_params.argtypes = [ctypes.c_void_p, _dp]
        L.pcc_oracle_cur_time.restype = ctypes.c_double
        L.pcc_oracle_cur_time.argtypes = [ctypes.c_void_p]
        L.pcc_oracle_heap_len.restype = ctypes.c_long
        L.pcc_oracle_heap_len.argtypes = [ctypes.c_void_p]
        L.pcc_oracle_link_state.restype = ctypes.c_double
        L.pcc_oracle_link_state.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.pcc_oracle_np_mean.restype = ctypes.c_double
        L.pcc_oracle_np_mean.argtypes = [_dp, ctypes.c_long]
        L.pcc_oracle_philox.argtypes = [_u32p, _u32p, _u32p]
        L.pcc_oracle_metric_table.argtypes = [_dp, _dp, _dp]
        L.pcc_oracle_mt_fill.argtypes = [ctypes.c_uint64, ctypes.c_long, _dp, ctypes.c_long]
        L.pcc_oracle_use_cwnd.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.pcc_oracle_cwnd.restype = ctypes.c_long
        L.pcc_oracle_cwnd.argtypes = [ctypes.c_void_p, ctypes.c_int]
        L.pcc_oracle_apply_cwnd_actions.argtypes = [ctypes.c_void_p, _dp, ctypes.c_double]
        L.pcc_oracle_run_batch.restype = ctypes.c_int
        L.pcc_oracle_run_batch.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int,
            ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, _u64p,
            ctypes.c_long, _dp, ctypes.c_long, _dp, _dp, _dp, _dp, _dp, _dp, _dp, ctypes.c_int]
        L.pcc_oracle_run_batch_cwnd.restype = ctypes.c_int
        L.pcc_oracle_run_batch_cwnd.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int,
            ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, _u64p,
            ctypes.c_long, _dp, ctypes.c_long, _dp, _dp, _dp, _dp, _dp, _dp, _dp, _dp, ctypes.c_int]
        L.pcc_oracle_run_batch_opts.restype = ctypes.c_int
        L.pcc_oracle_run_batch_opts.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int,
            ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, _u64p,
            ctypes.c_long, _dp, ctypes.c_long, _dp, _dp, _dp, ctypes.c_double, _dp, _dp, _dp, _dp, _dp, ctypes.c_int]
        L.pcc_oracle_use_latency_noise.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_double]
        L.pcc_oracle_run_batch_at.restype = ctypes.c_int
        L.pcc_oracle_run_batch_at.argtypes = [
            ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, _ip, ctypes.c_int,
            ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_uint64, ctypes.c_uint32, _u64p,
            ctypes.c_long, _dp, ctypes.c_long, _dp, _dp, _dp, ctypes.c_double, _ip, _dp, _dp, _dp, _dp, _dp, ctypes.c_int]
        _lib = L
    return _lib


def _ptr(a, typ=_dp):
    return None if a is None else a.ctypes.data_as(typ)


def np_mean(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return lib().pcc_oracle_np_mean(_ptr(a), a.size)


def philox4x32(ctr, key):
    c = np.asarray(ctr, dtype=np.uint32)
    k = np.asarray(key, dtype=np.uint32)
    o = np.zeros(4, dtype=np.uint32)
    lib().pcc_oracle_philox(_ptr(c, _u32p), _ptr(k, _u32p), _ptr(o, _u32p))
    return o


def mt_uniforms(seed, n, skip=0):
    """First n values of random.Random(seed).random() after `skip` draws."""
    out = np.empty(n, dtype=np.float64)
    lib().pcc_oracle_mt_fill(int(seed), int(skip), _ptr(out), int(n))
    return out


def metric_table():
    mn, mx, sc = (np.zeros(N_METRICS) for _ in range(3))
    lib().pcc_oracle_metric_table(_ptr(mn), _ptr(mx), _ptr(sc))
    return mn, mx, sc


class OracleEnv(object):
    """One env object with the reference's life cycle (constructor / reset / step)."""

    def __init__(self, n_senders=1, history_len=10, features=DEFAULT_FEATURES,
                 mean_mode=MEAN_NUMPY, delta_scale=0.025):
        self.L = lib()
        self.n_senders, self.history_len = n_senders, history_len
        self.fids = np.asarray(feature_ids(features), dtype=np.int32)
        self.HF = history_len * len(self.fids)
        self.delta_scale = delta_scale
        self.h = self.L.pcc_oracle_create(n_senders, history_len, _ptr(self.fids, _ip), len(self.fids), mean_mode)
        if not self.h:
            raise ValueError("pcc_oracle_create rejected the arguments")
        self._keep = None

    def close(self):
        if getattr(self, "h", None):
            self.L.pcc_oracle_destroy(self.h)
            self.h = None

    __del__ = close

    def rng_mt(self, seed, skip=0):
        self.L.pcc_oracle_rng_mt(self.h, int(seed))
        if skip:
            self.L.pcc_oracle_rng_skip(self.h, int(skip))

    def rng_trace(self, u):
        self._keep = np.ascontiguousarray(u, dtype=np.float64)
        self.L.pcc_oracle_rng_trace(self.h, _ptr(self._keep), self._keep.size)

    def rng_philox(self, seed, env_gid):
        self.L.pcc_oracle_rng_philox(self.h, int(seed), int(env_gid))

    def set_params(self, bw, dl, queue, loss, rate0):
        r = np.atleast_1d(np.asarray(rate0, dtype=np.float64))
        self.L.pcc_oracle_set_params(self.h, bw, dl, float(queue), loss, _ptr(r))

    def reset(self):
        obs = np.zeros((self.n_senders, self.HF))
        self.L.pcc_oracle_reset(self.h, _ptr(obs))
        return obs[0] if self.n_senders == 1 else obs

    def use_cwnd(self, on=True):
        """The reference's dormant USE_CWND engine option (ns:54): window-limited sending, and
        step() takes [rate action, cwnd action] (ns:376-377, 413-414)."""
        self._cwnd = bool(on)
        self.L.pcc_oracle_use_cwnd(self.h, 1 if on else 0)

    def use_latency_noise(self, on=True, max_noise=1.1):
        """The reference's dormant USE_LATENCY_NOISE engine option (ns:51-52, 150-151, 171-172): every
        link latency is multiplied by random.uniform(1.0, MAX_LATENCY_NOISE), packets can overtake."""
        self.L.pcc_oracle_use_latency_noise(self.h, 1 if on else 0, float(max_noise))

    def cwnd(self, sender=0):
        return int(self.L.pcc_oracle_cwnd(self.h, sender))

    def step(self, action):
        a = np.atleast_1d(np.asarray(action, dtype=np.float64))
        if getattr(self, "_cwnd", False):
            a = a.reshape(self.n_senders, 2) if a.size == 2 * self.n_senders else a.reshape(1, 2)
            c = np.ascontiguousarray(a[:, 1])
            self.L.pcc_oracle_apply_cwnd_actions(self.h, _ptr(c), self.delta_scale)
            a = np.ascontiguousarray(a[:, 0])
        obs = np.zeros((self.n_senders, self.HF))
        rew = np.zeros(self.n_senders)
        row = np.zeros((self.n_senders, STEP_COLS))
        done = self.L.pcc_oracle_step(self.h, _ptr(a), self.delta_scale, _ptr(obs), _ptr(rew), _ptr(row))
        if done < 0:
            raise TypeError("step() before reset(): run_dur is None")
        self.last_row = row
        if self.n_senders == 1:
            return obs[0], float(rew[0]), bool(done), {}
        return obs, rew, bool(done), {}

    def params(self):
        out = np.zeros(5 + self.n_senders)
        self.L.pcc_oracle_get_params(self.h, _ptr(out))
        return out

    @property
    def cur_time(self):
        return self.L.pcc_oracle_cur_time(self.h)

    @property
    def heap_len(self):
        return self.L.pcc_oracle_heap_len(self.h)

    @property
    def rng_draws(self):
        return self.L.pcc_oracle_rng_draws(self.h)


def run_batch(actions, n_senders=1, history_len=10, features=DEFAULT_FEATURES, mean_mode=MEAN_NUMPY,
              delta_scale=0.025, rng_mode=RNG_PHILOX, seed=0, env_gid_base=0, mt_seeds=None, mt_skip=5,
              trace=None, params=None, n_episodes=1, n_threads=None, want_obs=True, cwnd_actions=None,
              latency_noise=None, first_episode=None):
    """Run B independent envs for T steps.  actions: [B, T] or [B, T, n_senders].
    cwnd_actions (same shape) switches the USE_CWND engine option on and supplies the second
    action component; latency_noise (e.g. 1.1 = the reference's MAX_LATENCY_NOISE) switches
    USE_LATENCY_NOISE on.  first_episode (an int or [B] ints, Philox uniforms only): the episode index env b starts at
    -- with Philox an episode's links and draws are keyed by (env id, episode index), so an env in its k-th episode can
    be checked without replaying the k episodes before it.

    Returns dict(steps [B, S, T, 19], obs [B, S, T, H*F], obs0 [B, S, H*F],
                 params [B, 5+S], warm [B, 2]); S axis squeezed when n_senders == 1.
    """
    a = np.ascontiguousarray(actions, dtype=np.float64)
    if a.ndim == 2:
        a = a[:, :, None]
    B, T, S = a.shape
    assert S == n_senders
    fids = np.asarray(feature_ids(features), dtype=np.int32)
    HF = history_len * len(fids)
    steps = np.zeros((B, S, T, STEP_COLS))
    obs = np.zeros((B, S, T, HF)) if want_obs else None
    obs0 = np.zeros((B, S, HF))
    pout = np.zeros((B, 5 + S))
    warm = np.zeros((B, 2))
    p = None if params is None else np.ascontiguousarray(params, dtype=np.float64).reshape(B, 4 + S)
    tr = None if trace is None else np.ascontiguousarray(trace, dtype=np.float64).reshape(B, -1)
    ms = None if mt_seeds is None else np.ascontiguousarray(mt_seeds, dtype=np.uint64)
    nt = n_threads if n_threads else (os.cpu_count() or 1)
    ca = None if cwnd_actions is None else np.ascontiguousarray(cwnd_actions, dtype=np.float64).reshape(B, T, S)
    fe = None
    if first_episode is not None:
        if rng_mode != RNG_PHILOX:
            raise ValueError("first_episode needs Philox uniforms (the other streams depend on the episodes before)")
        fe = np.ascontiguousarray(np.broadcast_to(np.asarray(first_episode, dtype=np.int32), (B,)))
    bad = lib().pcc_oracle_run_batch_at(
        B, S, T, n_episodes, history_len, _ptr(fids, _ip), len(fids), mean_mode, delta_scale, rng_mode,
        int(seed), int(env_gid_base), _ptr(ms, _u64p), int(mt_skip), _ptr(tr),
        0 if tr is None else tr.shape[1], _ptr(p), _ptr(a), _ptr(ca), float(latency_noise or 0.0), _ptr(fe, _ip), _ptr(steps),
        _ptr(obs), _ptr(obs0), _ptr(pout), _ptr(warm), nt)
    if bad:
        raise RuntimeError("loss-uniform trace ran out for env %d" % (bad - 1))
    out = dict(steps=steps, obs=obs, obs0=obs0, params=pout, warm=warm)
    if S == 1:
        out["steps"] = steps[:, 0]
        out["obs"] = None if obs is None else obs[:, 0]
        out["obs0"] = obs0[:, 0]
    return out
