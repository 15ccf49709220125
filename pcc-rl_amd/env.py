"""Env surface of the MI355X-native simulator.

Two classes:

* ``BatchedNetworkEnv`` -- N independent envs advanced in lockstep on one GPU, torch tensors
  in and out, no host round trips in ``step``.  This is the fast path.
* ``SimulatedNetworkEnv`` -- the reference's old-gym single-env protocol
  (src/gym/network_sim.py:344-496: ``reset() -> obs``, ``step(a) -> (obs, reward, done, {})``,
  ``seed``, ``render``, ``close``, ``observation_space``, ``action_space``, constructor
  arguments ``history_len`` and ``features``) as a view onto a batch of one, so agent code
  written against the reference runs unchanged.

All simulation runs in the HIP library behind include/pcc_sim.h; this file only owns the
output tensors and argument checking.  There is no CPU implementation to fall back to.
"""
import ctypes

import numpy as np
import torch

from . import native
from .config import DELTA_SCALE, arg_or_default
from .metrics import DEFAULT_FEATURES, feature_ids, get_max_obs_vector, get_min_obs_vector, metric_info
from .native import PccError, check, lib
from .spaces import Box

MAX_STEPS = 400          # ns:41
DEFAULT_RING_CAPACITY = 32768

_TORCH_DTYPES = {"float64": torch.float64, "int32": torch.int32, "int64": torch.int64, "uint8": torch.uint8}


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


class BatchedNetworkEnv(object):
    """N single-bottleneck congestion-control envs on one MI355X.

    Parameters mirror the reference constructor (``history_len``, ``features``) plus what a
    batch needs.  ``link_params`` = None draws (bw, latency, queue, loss, rate0) per episode
    from the reference's ranges (ns:355-358, 455-466); otherwise a dict/tuple of five tensors
    or scalars ``(bw, dl, queue, loss, rate0)`` fixes them.

    Shapes (S = n_senders): actions ``[N]``/``[N, 1]`` (S=1) or ``[N, 2]``; obs ``[N, H*F]``
    (S=1) or ``[N, S, H*F]`` float32, oldest monitor interval first; reward ``[N]`` or
    ``[N, S]`` float32; done ``[N]`` bool.  With ``auto_reset`` (default) an env that reaches
    ``max_steps`` is reset inside the same ``step`` call and its obs row is the first
    observation of the next episode; ``episode_returns()`` holds the finished return per env.

    The tensors returned by ``reset``/``step`` are the env's own output buffers: they are
    overwritten by the next call (pass ``new_tensors=True`` to get fresh ones each call).
    """

    # batches below this many envs are stepped without work lists (None: the library's default, 8192; the GPU tests set
    # 0 here so that small batches exercise the work-list paths too)
    DEFAULT_LIST_MIN_ENVS = None
    # the fused step on / off and its acquire mode (None: the library's defaults; tests/conftest.py sets them from
    # PCC_TEST_FUSED / PCC_TEST_FUSED_ACQUIRE so that the parity suite can run through either path)
    DEFAULT_FUSED = None
    DEFAULT_FUSED_ACQUIRE = None
    DEFAULT_NOISE_SORTED = None   # tests: PCC_TUNE_NOISE_SORTED for every handle (0 = the event loop, 2 = the two crossed)

    def __init__(self, n_envs, device="cuda", history_len=None, features=None, seed=0, n_senders=1,
                 link_params=None, env_gid_base=0, ring_capacity=0, auto_reset=True, delta_scale=None,
                 max_steps=MAX_STEPS, record_steps=False, new_tensors=False, use_cwnd=False, latency_noise=None,
                 ring_pools=None):
        if history_len is None:
            history_len = arg_or_default("--history-len", default=10)
        if features is None:
            features = arg_or_default("--input-features", default=DEFAULT_FEATURES)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise ValueError("BatchedNetworkEnv runs on an MI355X only (device=%r); there is no CPU path" % (device,))
        if not torch.cuda.is_available():
            raise RuntimeError("no GPU visible: the simulator is a HIP library for gfx950 and has no CPU fallback")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.n_envs, self.n_senders = int(n_envs), int(n_senders)
        self.history_len = int(history_len)
        self.features = features.split(",") if isinstance(features, str) else list(features)
        self.feature_ids = feature_ids(self.features)
        self.obs_dim = self.history_len * len(self.feature_ids)
        self.auto_reset, self.record_steps, self.new_tensors = bool(auto_reset), bool(record_steps), bool(new_tensors)
        self.max_steps = int(max_steps)
        self.seed_value = int(seed)

        L = lib()
        fids = (ctypes.c_int32 * len(self.feature_ids))(*self.feature_ids)
        handle = ctypes.c_void_p()
        check(L.pcc_create(self.n_envs, self.n_senders, self.history_len, fids, len(self.feature_ids),
                           self.seed_value & (2 ** 64 - 1), int(env_gid_base), int(ring_capacity),
                           self.device.index, ctypes.byref(handle)))
        self._h = handle
        self._L = L
        if ring_pools is not None:
            # (div1, div2, div3): tier k of the in-flight ring pools holds a slot for one sender in div_k.  By default the
            # library sizes the pools from the free device memory (a slot for every sender where a third of it pays for that,
            # never less than divisors 2, 8, 32): a policy that saturates every link cannot run them dry
            check(L.pcc_set_ring_pools(self._h, *[int(v) for v in ring_pools]))
        if self.DEFAULT_LIST_MIN_ENVS is not None:
            check(L.pcc_set_tuning(self._h, 12, float(self.DEFAULT_LIST_MIN_ENVS)))
        if self.DEFAULT_FUSED is not None:
            check(L.pcc_set_tuning(self._h, 26, float(self.DEFAULT_FUSED)))
        if self.DEFAULT_FUSED_ACQUIRE is not None:
            check(L.pcc_set_tuning(self._h, 27, float(self.DEFAULT_FUSED_ACQUIRE)))
        if self.DEFAULT_NOISE_SORTED is not None:
            check(L.pcc_set_tuning(self._h, 33, float(self.DEFAULT_NOISE_SORTED)))
        check(L.pcc_set_delta_scale(self._h, float(DELTA_SCALE if delta_scale is None else delta_scale)))
        check(L.pcc_set_max_steps(self._h, self.max_steps))
        # the reference's dormant USE_CWND engine option (ns:54): window-limited sending, actions
        # become [rate action, cwnd action] (ns:376-377, 412-414)
        self.use_cwnd = bool(use_cwnd)
        self.action_dim = 2 if self.use_cwnd else 1
        if self.use_cwnd:
            check(L.pcc_set_cwnd_mode(self._h, 1))
        # ... and USE_LATENCY_NOISE (ns:51-52): latency_noise = MAX_LATENCY_NOISE (the reference: 1.1), None = off
        self.latency_noise = None if not latency_noise else float(latency_noise)
        if self.latency_noise:
            check(L.pcc_set_latency_noise(self._h, 1, self.latency_noise))

        N, S, D = self.n_envs, self.n_senders, self.obs_dim
        with torch.cuda.device(self.device):
            self._obs = torch.empty((N, S, D), dtype=torch.float32, device=self.device)
            self._reward = torch.empty((N, S), dtype=torch.float32, device=self.device)
            self._done = torch.empty((N,), dtype=torch.uint8, device=self.device)
            self._steps = (torch.empty((N, S, native.PCC_STEP_COLS), dtype=torch.float64, device=self.device)
                           if self.record_steps else None)
        self._params = None
        self._trace = None
        self._was_reset = False
        self._t = 0
        if link_params is not None:
            self.set_link_params(*(link_params.values() if isinstance(link_params, dict) else link_params))

        single = get_min_obs_vector(self.features), get_max_obs_vector(self.features)
        self.single_observation_space = Box(np.tile(single[0], self.history_len), np.tile(single[1], self.history_len),
                                            dtype=np.float32)                       # ns:382-388
        self.single_action_space = Box(np.array([-1e12] * self.action_dim), np.array([1e12] * self.action_dim),
                                       dtype=np.float32)                             # ns:376-379
        self.observation_space = self.single_observation_space
        self.action_space = self.single_action_space
        self.num_envs = self.n_envs

    # ------------------------------------------------------------------ configuration
    def _stream(self):
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _as_param(self, x, per_sender=False):
        shape = (self.n_senders, self.n_envs) if per_sender else (self.n_envs,)
        t = torch.as_tensor(x, dtype=torch.float64, device=self.device)
        if t.ndim == 0:
            t = t.expand(shape)
        elif per_sender and t.shape == (self.n_envs,) and self.n_senders == 1:
            t = t.reshape(shape)
        elif per_sender and t.shape == (self.n_envs, self.n_senders):
            t = t.t()
        if tuple(t.shape) != shape:
            raise ValueError("link parameter has shape %s, expected %s" % (tuple(t.shape), shape))
        return t.contiguous().clone()

    def set_link_params(self, bw, dl, queue, loss, rate0):
        """Fix (bandwidth pkt/s, one-way delay s, queue packets, loss prob, starting rate pkt/s)
        for every following reset (scalars or [N] tensors; rate0 [N] / [N, S])."""
        p = (self._as_param(bw), self._as_param(dl), self._as_param(queue), self._as_param(loss),
             self._as_param(rate0, per_sender=True))
        check(self._L.pcc_set_link_params(self._h, *[_ptr(t) for t in p]))
        self._params = p  # keep alive: the library reads them at reset time

    def randomize_link_params(self, ranges=None):
        """Back to per-episode random parameters; ``ranges`` optionally overrides
        {bw, lat, queue exponent, loss, rate0/bw} as ((lo...), (hi...))."""
        check(self._L.pcc_set_link_params(self._h, None, None, None, None, None))
        self._params = None
        if ranges is not None:
            lo = (ctypes.c_double * 5)(*[float(v) for v in ranges[0]])
            hi = (ctypes.c_double * 5)(*[float(v) for v in ranges[1]])
            check(self._L.pcc_set_param_ranges(self._h, lo, hi))

    def set_loss_trace(self, u):
        """Parity mode: replay per-packet loss uniforms u[env, k] (k = k-th packet sent in the
        episode) instead of Philox.  ``None`` switches back."""
        if u is None:
            check(self._L.pcc_set_rng(self._h, native.PCC_RNG_PHILOX, None, 0))
            self._trace = None
            return
        t = torch.as_tensor(u, dtype=torch.float64, device=self.device).contiguous()
        if t.ndim != 2 or t.shape[0] != self.n_envs:
            raise ValueError("loss trace must be [n_envs, K]")
        check(self._L.pcc_set_rng(self._h, native.PCC_RNG_TRACE, _ptr(t), t.shape[1]))
        self._trace = t

    def set_tuning(self, round_packets=None, takeover_lanes=None, send_envs_per_wave=None, heavy_predict=None,
                   send_waves=None, team_predict=None, heavy_item_packets=None, retire_wide_predict=None, list_min_envs=None,
                   retire_sorted=None, light_snake=None, wave_oldest_first=None, prio_level=None, prio_light_items=None,
                   prio_wave_items=None, prio_team=None, retire_grid_frac=None, restart_fork=None, parts=None, light_half_predict=None,
                   fused=None, fused_acquire=None, fused_light_wgs=None, fused_max_naps=None, fused_partial_naps=None, fused_debug=None, fused_light_front=None,
                   noise_sorted=None, light_wgs=None, light_front=None):
        """Performance knobs (results do not depend on them); see pcc_set_tuning in include/pcc_sim.h.  `parts` (the
        partitioning of the batch) must be followed by reset()."""
        for key, value in ((2, round_packets), (3, takeover_lanes), (4, send_envs_per_wave), (5, heavy_predict),
                           (8, send_waves), (9, team_predict), (10, heavy_item_packets), (11, retire_wide_predict), (12, list_min_envs),
                           (13, retire_sorted), (14, light_snake), (15, wave_oldest_first), (16, prio_level), (17, prio_light_items),
                           (18, prio_wave_items), (19, prio_team), (22, retire_grid_frac), (23, restart_fork), (24, parts), (25, light_half_predict),
                           (26, fused), (27, fused_acquire), (28, fused_light_wgs), (29, fused_max_naps), (30, fused_partial_naps), (31, fused_debug), (32, fused_light_front),
                           (33, noise_sorted), (34, light_wgs), (35, light_front)):
            if value is not None:
                check(self._L.pcc_set_tuning(self._h, key, float(value)))

    def seed(self, seed=None):
        if seed is not None:
            self.seed_value = int(seed)
            check(self._L.pcc_set_seed(self._h, self.seed_value & (2 ** 64 - 1)))
        return [seed]

    # ------------------------------------------------------------------ protocol
    def _out(self, t):
        if t is None:
            return None
        t = t.clone() if self.new_tensors else t
        return t[:, 0] if self.n_senders == 1 else t

    def reset(self, mask=None):
        """Reset all envs (or those where ``mask`` is true); returns the observation tensor."""
        m = None
        if mask is not None:
            m = torch.as_tensor(mask, device=self.device).to(torch.uint8).contiguous()
            if m.shape != (self.n_envs,):
                raise ValueError("mask must be [n_envs]")
        check(self._L.pcc_reset(self._h, _ptr(m), _ptr(self._obs), self._stream()))
        self._was_reset = True
        if mask is None:
            self._t = 0
        return self._out(self._obs)

    def _actions(self, actions):
        a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions), device=self.device)
        if a.device != self.device:
            a = a.to(self.device)
        if a.dtype not in (torch.float32, torch.float64):
            a = a.to(torch.float32)
        width = self.n_senders * self.action_dim
        if a.numel() != self.n_envs * width:
            raise ValueError("actions has %d elements, expected n_envs * %d = %d"
                             % (a.numel(), width, self.n_envs * width))
        return a.reshape(self.n_envs, width).contiguous()

    def step_send(self, actions):
        """First half of step(): apply the actions and transmit the coming monitor interval's packets."""
        a = self._actions(actions)
        check(self._L.pcc_step_send(self._h, _ptr(a), 1 if a.dtype == torch.float64 else 0, self._stream()))

    def step_retire(self):
        """Second half of step(): acknowledgements, losses, metrics; returns what step() returns."""
        check(self._L.pcc_step_retire(self._h, _ptr(self._obs), _ptr(self._reward), _ptr(self._done),
                                      _ptr(self._steps), 1 if self.auto_reset else 0, self._stream()))
        return self._step_result()

    def _step_result(self):
        self._t += 1
        info = {}
        if self._steps is not None:
            info["steps"] = self._out(self._steps)
        done = self._done.clone() if self.new_tensors else self._done
        return self._out(self._obs), self._out(self._reward), done.view(torch.bool), info

    def step(self, actions):
        """One monitor interval for every env (ns:407-446 batched): obs, reward, done, info.  One
        library call (pcc_step: two launches, the send half and the retire half)."""
        a = self._actions(actions)
        check(self._L.pcc_step(self._h, _ptr(a), 1 if a.dtype == torch.float64 else 0, _ptr(self._obs),
                               _ptr(self._reward), _ptr(self._done), _ptr(self._steps),
                               1 if self.auto_reset else 0, self._stream()))
        return self._step_result()

    def step_into(self, actions, obs_out, reward_out, done_out):
        """step() that writes straight into the caller's tensors -- e.g. rows of a rollout buffer -- instead of the
        env's own output buffers: obs_out float32 [N, S, H*F] (or [N, H*F] with one sender), reward_out float32 [N, S]
        ([N]), done_out uint8/bool [N], all contiguous on the env's device.  Returns nothing; no copy is made."""
        a = self._actions(actions)
        N, S, D = self.n_envs, self.n_senders, self.obs_dim
        for t, n, dt in ((obs_out, N * S * D, (torch.float32,)), (reward_out, N * S, (torch.float32,)),
                         (done_out, N, (torch.uint8, torch.bool))):
            if t.numel() != n or t.dtype not in dt or not t.is_contiguous() or t.device != self.device:
                raise ValueError("step_into: an output tensor has the wrong size, dtype, layout or device")
        check(self._L.pcc_step(self._h, _ptr(a), 1 if a.dtype == torch.float64 else 0, _ptr(obs_out), _ptr(reward_out),
                               _ptr(done_out), _ptr(self._steps), 1 if self.auto_reset else 0, self._stream()))
        self._t += 1

    def step_many(self, actions, obs_out=None, reward_out=None, done_out=None, steps_out=None):
        """T steps with pre-computed actions in ONE library call (pcc_step_many): actions [T, N(, S)(, 2)]; the optional outputs
        [T, N, S, H*F] / [T, N, S] / [T, N] / [T, N, S, 19] float64 (contiguous, on the env's device) take every step's
        observation, reward, done row and full-precision step record (``record_steps`` does not apply here: pass steps_out).
        For open-loop drivers of small batches: below ``list_min_envs`` envs, in lockstep, the steps up to the next episode
        boundary run inside one launch.  If the call fails part-way the error says after how many steps; the env's step
        counter is not advanced then (reset before going on)."""
        a = actions if torch.is_tensor(actions) else torch.as_tensor(np.asarray(actions), device=self.device)
        if a.device != self.device:
            a = a.to(self.device)
        if a.dtype not in (torch.float32, torch.float64):
            a = a.to(torch.float32)
        T = int(a.shape[0])
        width = self.n_senders * self.action_dim
        if T < 1 or a.numel() != T * self.n_envs * width:
            raise ValueError("actions has %d elements, expected T * n_envs * %d" % (a.numel(), width))
        a = a.reshape(T, self.n_envs * width).contiguous()
        N, S, D = self.n_envs, self.n_senders, self.obs_dim
        for t, n, dt in ((obs_out, T * N * S * D, (torch.float32,)), (reward_out, T * N * S, (torch.float32,)),
                         (done_out, T * N, (torch.uint8, torch.bool)), (steps_out, T * N * S * native.PCC_STEP_COLS, (torch.float64,))):
            if t is not None and (t.numel() != n or t.dtype not in dt or not t.is_contiguous() or t.device != self.device):
                raise ValueError("step_many: an output tensor has the wrong size, dtype, layout or device")
        check(self._L.pcc_step_many(self._h, _ptr(a), 1 if a.dtype == torch.float64 else 0, T, _ptr(obs_out), _ptr(reward_out),
                                    _ptr(done_out), _ptr(steps_out), 1 if self.auto_reset else 0, self._stream()))
        self._t += T

    # ------------------------------------------------------------------ introspection
    def state(self, name):
        """Copy of one internal state field as a tensor (see native.FIELDS)."""
        fid, dtype, per_sender = native.FIELDS[name]
        shape = (self.n_senders, self.n_envs) if per_sender else (self.n_envs,)
        out = torch.empty(shape, dtype=_TORCH_DTYPES[dtype], device=self.device)
        check(self._L.pcc_get_state(self._h, fid, _ptr(out), self._stream()))
        return out

    def restart_stats(self):
        """How episodes of envs out of lockstep were started so far: {"shadow_swaps": ..., "restart_list": ...}
        (pcc_restart_stats); synchronizes the env's stream."""
        out = (ctypes.c_uint64 * 2)()
        check(self._L.pcc_restart_stats(self._h, out, self._stream()))
        return {"shadow_swaps": int(out[0]), "restart_list": int(out[1])}

    def debug_addresses(self):
        """Device addresses of the handle's allocations (pcc_debug_addresses): state, ring tiers 0..3, lists, shadow rings, history."""
        out = (ctypes.c_uint64 * 8)()
        check(self._L.pcc_debug_addresses(self._h, out))
        return dict(zip(("state", "tier0", "tier1", "tier2", "tier3", "lists", "shadow_rings", "hist"), [int(v) for v in out]))

    def fused_steps(self):
        """Steps of this handle that ran as ONE launch so far (pcc_fused_steps; set_tuning(fused=0) switches that off)."""
        out = ctypes.c_uint64(0)
        check(self._L.pcc_fused_steps(self._h, ctypes.byref(out)))
        return int(out.value)

    def episode_returns(self):
        """Return of the last finished episode per env ([N] or [S, N]), float64."""
        r = self.state("last_return")
        return r[0] if self.n_senders == 1 else r

    def check_flags(self):
        """Raise if any env overflowed its in-flight ring, ran out of loss trace, found the ring pools empty
        or was given link parameters outside what the simulator covers."""
        flags = self.state("flags")
        bad = int((flags != 0).sum().item())
        if bad:
            count = lambda bit: int(((flags & bit) != 0).sum().item())
            raise PccError(-6, "%d envs overflowed the in-flight ring, %d ran past the loss trace, %d found the ring pools "
                               "empty (ring_pools), %d have link parameters out of range, %d were given a NaN action, "
                               "%d ran their clock out of the supported range, %d hit an internal error"
                           % (count(native.PCC_FLAG_RING_OVERFLOW), count(native.PCC_FLAG_TRACE_OVERRUN),
                              count(native.PCC_FLAG_POOL_EXHAUSTED), count(native.PCC_FLAG_BAD_PARAMS),
                              count(native.PCC_FLAG_BAD_ACTION), count(native.PCC_FLAG_TIME_RANGE),
                              count(native.PCC_FLAG_INTERNAL)))

    @property
    def device_bytes(self):
        return int(self._L.pcc_device_bytes(self._h))

    def debug_timeline(self):
        """[send work items (8 words each) ..., retire workgroups (2 rows of 8 each)] uint64 of the last
        step (PCC_DEBUG_TIMELINE=1 at creation; see pcc_debug_timeline in include/pcc_sim.h), or None
        when the timeline is off."""
        import numpy as np
        n = int(self._L.pcc_debug_timeline(self._h, None, 0))
        if n <= 0:
            return None
        out = np.zeros(n, dtype=np.uint64)
        got = int(self._L.pcc_debug_timeline(self._h, out.ctypes.data, n))
        return out[:got].reshape(-1, 8)

    PASS_STAT_NAMES = ("pass_empty", "pass_scan", "pass_free", "pass_serial", "pk_empty", "pk_scan", "pk_free",
                       "pk_serial", "scan_nothing_committed", "refused_time", "refused_queue", "envs_heavy_wave",
                       "envs_taken_over", "cycles_committed", "cycles_serial", "items")

    def debug_pass_stats(self, reset=True):
        """Counters of the send half's wave passes (PCC_DEBUG_TIMELINE=1 at creation), as a dict."""
        import numpy as np
        out = np.zeros(16, dtype=np.uint64)
        check(self._L.pcc_debug_pass_stats(self._h, out.ctypes.data, 1 if reset else 0))
        return dict(zip(self.PASS_STAT_NAMES, [int(v) for v in out]))

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._L.pcc_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def render(self, mode="human"):
        pass


class GroupedNetworkEnv(object):
    """N envs as G independent groups, each a ``BatchedNetworkEnv`` on its own HIP stream.

    The step time of one big batch is set by its slowest wavefront (a few envs with thousands of
    packets) while most of the GPU idles.  Groups that are stepped independently -- no per-step
    synchronization between them -- drift apart, and the bulk of one group fills the tail of another:
    the pattern of double-buffered sampling, where the policy runs on one group's observations
    while the other groups simulate.  Results are the same numbers as one ``BatchedNetworkEnv`` of
    ``n_envs`` envs (group g holds the global env ids ``g * n_envs / G ...``); only the schedule
    differs.  65 536 envs on one MI355X (round 3, ``bench.py --groups 2``): one batch 2.92e8, 2 groups 3.13e8
    env-steps/s; more groups are slower (a half-size launch is as long as a full one: the gain is only what one
    group's tail leaves to the other) -- and only when HIP maps the groups' streams to different hardware queues;
    streams that share a queue serialize, and the mapping is not under the caller's control.

    ``step_group(g, actions)`` enqueues group g's step on its stream and returns its (obs, reward,
    done, info) tensors, valid on that stream (``self.streams[g]``); ``synchronize()`` waits for all.
    """

    def __init__(self, n_envs, n_groups=2, device="cuda", seed=0, env_gid_base=0, **kwargs):
        if n_envs % n_groups:
            raise ValueError("n_envs must be a multiple of n_groups")
        self.n_envs, self.n_groups, self.group_size = int(n_envs), int(n_groups), int(n_envs) // int(n_groups)
        self.device = torch.device(device)
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(self.n_groups)]
        self.groups = []
        for g in range(self.n_groups):
            with torch.cuda.stream(self.streams[g]):
                self.groups.append(BatchedNetworkEnv(self.group_size, device=self.device, seed=seed,
                                                     env_gid_base=int(env_gid_base) + g * self.group_size, **kwargs))

    def reset_group(self, g, mask=None):
        with torch.cuda.stream(self.streams[g]):
            return self.groups[g].reset(mask)

    def step_group(self, g, actions):
        with torch.cuda.stream(self.streams[g]):
            return self.groups[g].step(actions)

    def reset(self):
        """Reset every group; returns the observations [n_envs, ...] (synchronizes)."""
        obs = [self.reset_group(g) for g in range(self.n_groups)]
        self.synchronize()
        return torch.cat(obs, 0)

    def synchronize(self):
        for s in self.streams:
            s.synchronize()

    def check_flags(self):
        for e in self.groups:
            e.check_flags()

    def close(self):
        for e in self.groups:
            e.close()


class SimulatedNetworkEnv(object):
    """Drop-in for the reference's ``SimulatedNetworkEnv`` (src/gym/network_sim.py:344-496).

    Same constructor arguments, spaces, ``reset``/``step``/``seed``/``render``/``close``
    signatures and return types (numpy observation of shape ``(history_len * n_features,)``,
    Python float reward, bool done, empty info dict).  One env = a batch of one on the GPU, so
    this adapter is for compatibility, not speed: use ``BatchedNetworkEnv`` to train.

    Observations are float64 arrays, as the reference's are at run time (``SenderHistory.as_array`` concatenates float64
    rows although the ``Box`` says float32: src/common/sender_obs.py:68-73): the adapter keeps the
    history itself, from the full-precision per-step record of the library (every metric of the new monitor interval
    divided by its scale, so:44-54), so an agent sees the reference's numbers bit for bit.  (``BatchedNetworkEnv``, the
    fast path, hands out the float32 rows the device writes -- the ``Box`` dtype.)

    Differences from the reference, all deliberate: ``seed()`` really seeds the simulator (the
    reference's creates an RNG nothing reads, ns:396-398); nothing is printed; the JSON event
    log is written only when ``dump_events_to_file`` is called (the reference writes
    ``pcc_env_log_run_N.json`` into the CWD every 100 episodes, ns:475-476).
    """

    metadata = {"render.modes": []}

    def __init__(self, history_len=None, features=None, device="cuda", seed=0, link_params=None, use_cwnd=False,
                 latency_noise=None):
        self._env = BatchedNetworkEnv(1, device=device, history_len=history_len, features=features, seed=seed,
                                      link_params=link_params, auto_reset=False, record_steps=True,
                                      use_cwnd=use_cwnd, latency_noise=latency_noise)
        self.history_len = self._env.history_len
        self.features = self._env.features
        self.observation_space = self._env.single_observation_space
        self.action_space = self._env.single_action_space
        self.max_steps = MAX_STEPS
        self.steps_taken = 0
        self.reward_sum = 0.0
        self.reward_ewma = 0.0
        self.episodes_run = -1
        self.event_record = {"Events": []}
        self.run_dur = None
        self.viewer = None
        self._cols = [native.STEP_COLUMNS.index(f) for f in self.features]
        self._scales = np.array([metric_info(f)[2] for f in self.features], dtype=np.float64)
        self._hist = None

    def seed(self, seed=None):
        self._env.seed(seed)
        return [seed]

    def _obs64(self):
        return np.concatenate(self._hist)                                     # so:68-73: oldest interval first

    def reset(self):
        self._env.reset()
        # so:57-62: a history of empty intervals -- every metric of one is 0 but the two ratios (so:179-191)
        empty = np.array([1.0 if f in ("send ratio", "latency ratio") else 0.0 for f in self.features]) / self._scales
        self._hist = [empty.copy() for _ in range(self.history_len)]
        self.steps_taken = 0
        self.episodes_run += 1
        self.event_record = {"Events": []}
        self.reward_ewma = 0.99 * self.reward_ewma + 0.01 * self.reward_sum   # ns:480-481
        self.reward_sum = 0.0
        self.run_dur = float(self._env.state("run_dur")[0].item())
        return self._obs64()

    def step(self, actions):
        if self.run_dur is None:
            raise TypeError("step() called before reset(): run_dur is None")  # what ns:368,416 raises
        flat = np.asarray(actions, dtype=np.float64).reshape(-1)
        a = [float(flat[0])]                                                  # ns:409-412: action[0] ...
        if self._env.use_cwnd:
            a.append(float(flat[1]))                                          # ... and action[1] with USE_CWND (ns:413-414)
        act = torch.tensor([a], dtype=torch.float64, device=self._env.device)
        obs, reward, done, info = self._env.step(act)
        row = info["steps"][0].cpu().numpy()
        reward = float(row[native.STEP_COLUMNS.index("reward")])
        self.steps_taken += 1
        col = native.STEP_COLUMNS.index
        self.event_record["Events"].append({                                   # ns:422-436
            "Name": "Step", "Time": self.steps_taken, "Reward": reward,
            "Send Rate": float(row[col("send rate")]), "Throughput": float(row[col("recv rate")]),
            "Latency": float(row[col("avg latency")]), "Loss Rate": float(row[col("loss ratio")]),
            "Latency Inflation": float(row[col("sent latency inflation")]),
            "Latency Ratio": float(row[col("latency ratio")]), "Send Ratio": float(row[col("send ratio")])})
        self.run_dur = float(row[col("run_dur")])
        self.reward_sum += reward
        self._hist.pop(0)                                                      # so:64-66
        self._hist.append(row[self._cols] / self._scales)
        return self._obs64(), reward, bool(done[0].item()), {}

    def render(self, mode="human"):
        pass

    def close(self):
        self._env.close()

    def dump_events_to_file(self, filename):
        import json
        with open(filename, "w") as f:
            json.dump(self.event_record, f, indent=4)


def make(env_id="PccNs-v0", **kwargs):
    """``gym.make('PccNs-v0')`` without gym (ns:498 registers that id)."""
    if env_id != "PccNs-v0":
        raise KeyError("unknown env id %r (only 'PccNs-v0' is provided)" % (env_id,))
    return SimulatedNetworkEnv(**kwargs)


def register_gym():
    """Register 'PccNs-v0' with gym/gymnasium when one of them is installed (ns:498)."""
    done = []
    for mod in ("gymnasium", "gym"):
        try:
            reg = __import__(mod + ".envs.registration", fromlist=["register"]).register
            reg(id="PccNs-v0", entry_point="pcc_rl_amd:SimulatedNetworkEnv")
            done.append(mod)
        except Exception:
            continue
    return done
