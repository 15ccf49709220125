"""Whole-batch, whole-episode parity at the sizes bench.py runs (BASELINE.json configs[2] and configs[4]):
EVERY env of the batch, EVERY step of a 400-step episode and the first steps of the next one after the
auto-reset, all 19 step columns and the observation rows bit-equal to the CPU oracle -- not a sampled sliver.
The envs that take the exotic paths at full size (10-13 k-packet giants late in an episode on team items, the
top ring tiers, queue limits that straddle a power of two, envs promoted mid-episode) are thereby compared at
the size the bench runs them, wherever the work lists put them.

The oracle side runs on all host cores (OpenMP over envs, about 80 s for 65 536 x 400 env-steps on the GPU box);
the comparison goes over the envs in chunks so that the host never holds more than one chunk of oracle output.
PCC_FULL_SIZE_ENVS / PCC_FULL_SIZE_STEPS shrink the batch / the episode for quick runs (default: the full size).

ns = src/gym/network_sim.py of the reference: the whole of SimulatedNetworkEnv.step / reset (ns:406-484) is
what is compared, through the C ABI (pcc_step) on one side and oracle/pcc_oracle.c on the other."""
import os

import numpy as np
import pytest
import torch

import oracle
import pcc_rl_amd

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(int(os.environ.get("PCC_FULL_SIZE_TIMEOUT", "1500")), method="thread")]
DEV = "cuda:0"
CHUNK = 8192


def _first_mismatch(got, want, what):
    """Names the first differing (env, sender?, step, column): a failure at this size must say where to look."""
    if got.shape != want.shape:
        return "%s: shape %s vs %s" % (what, got.shape, want.shape)
    neq = ~((got == want) | (np.isnan(got) & np.isnan(want)))
    if not neq.any():
        return None
    idx = np.argwhere(neq)
    first = tuple(int(v) for v in idx[0])
    return "%s: %d of %d values differ, first at %s: got %r, oracle %r (envs affected: %d)" % (
        what, int(neq.sum()), neq.size, first, got[first], want[first], len(np.unique(idx[:, 0])))


def _run_and_compare(n_envs, n_senders, seed, n_steps, extra_steps, max_steps=None, tuning=None):
    """One whole episode of `n_steps` (= the env's max_steps) plus `extra_steps` of the next one with auto-reset, every
    step recorded on the device; then chunk by chunk against the oracle."""
    T = n_steps + extra_steps
    S = n_senders
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, n_senders=S, record_steps=True, auto_reset=True,
                                       max_steps=n_steps)
    if tuning:
        env.set_tuning(**tuning)
    obs0 = env.reset().clone()
    gen = torch.Generator(device=DEV).manual_seed(1234)
    acts = torch.rand((T, n_envs, S), generator=gen, device=DEV, dtype=torch.float32) * 2 - 1   # bench.py's action law
    rows = torch.empty((T, n_envs, S, 19), dtype=torch.float64, device=DEV)
    obs = torch.empty((T, n_envs, S, env.obs_dim), dtype=torch.float32, device=DEV)
    done = torch.empty((T, n_envs), dtype=torch.bool, device=DEV)
    for t in range(T):
        o, r, d, info = env.step(acts[t])
        rows[t] = info["steps"].reshape(n_envs, S, 19)
        obs[t] = o.reshape(n_envs, S, -1)
        done[t] = d
    torch.cuda.synchronize()
    env.check_flags()
    assert bool(done[n_steps - 1].all()) and int(done.sum().item()) == n_envs, "every env finishes exactly once, at step %d" % n_steps
    ret = env.episode_returns().cpu().numpy().reshape(S, n_envs) if S > 1 else env.episode_returns().cpu().numpy().reshape(1, n_envs)
    tiers = env.state("ring_tier").cpu().numpy()
    packets = float(rows[..., 0].sum().item())
    a_host = acts.to(torch.float64).permute(1, 0, 2).contiguous().cpu().numpy()   # [N, T, S]
    problems = []
    for lo in range(0, n_envs, CHUNK):
        hi = min(n_envs, lo + CHUNK)
        a = a_host[lo:hi]
        got_rows = rows[:, lo:hi].permute(1, 2, 0, 3).contiguous().cpu().numpy()   # [n, S, T, 19]
        got_obs = obs[:, lo:hi].permute(1, 2, 0, 3).contiguous().cpu().numpy()     # [n, S, T, HF]
        ref = oracle.run_batch(a[:, :n_steps], n_senders=S, rng_mode=oracle.RNG_PHILOX, seed=seed, env_gid_base=lo)
        ref_steps = ref["steps"].reshape(hi - lo, S, n_steps, 19)
        ref_obs = ref["obs"].reshape(hi - lo, S, n_steps, -1).astype(np.float32)
        ref_obs0 = ref["obs0"].reshape(hi - lo, S, -1).astype(np.float32)
        problems.append(_first_mismatch(obs0[lo:hi].reshape(hi - lo, S, -1).cpu().numpy(), ref_obs0, "envs %d..: reset observation" % lo))
        problems.append(_first_mismatch(got_rows[:, :, :n_steps, :3], ref_steps[..., :3], "envs %d..: sent/acked/lost" % lo))
        problems.append(_first_mismatch(got_rows[:, :, :n_steps], ref_steps, "envs %d..: step columns" % lo))
        # the observation of the episode's last step is the first one of the next episode (auto-reset): checked below
        problems.append(_first_mismatch(got_obs[:, :, :n_steps - 1], ref_obs[:, :, :n_steps - 1], "envs %d..: observations" % lo))
        # (the return is a running sum in step order; numpy sums pairwise: equal to rounding)
        assert np.allclose(ret[:, lo:hi].T, ref_steps[..., 6].sum(2), rtol=1e-9, atol=1e-9), "episode returns"
        if extra_steps:
            ref2 = oracle.run_batch(a[:, n_steps:], n_senders=S, rng_mode=oracle.RNG_PHILOX, seed=seed, env_gid_base=lo,
                                    first_episode=1)
            ref2_steps = ref2["steps"].reshape(hi - lo, S, extra_steps, 19)
            ref2_obs = ref2["obs"].reshape(hi - lo, S, extra_steps, -1).astype(np.float32)
            problems.append(_first_mismatch(got_obs[:, :, n_steps - 1], ref2["obs0"].reshape(hi - lo, S, -1).astype(np.float32),
                                            "envs %d..: first observation after the auto-reset" % lo))
            problems.append(_first_mismatch(got_rows[:, :, n_steps:], ref2_steps, "envs %d..: step columns of the second episode" % lo))
            problems.append(_first_mismatch(got_obs[:, :, n_steps:], ref2_obs, "envs %d..: observations of the second episode" % lo))
        problems = [p for p in problems if p]
        assert not problems, "\n".join(problems)
    env.close()
    return {"packets_per_env_step": packets / (n_envs * T), "max_packets_in_a_step": float(rows[..., 0].max().item()),
            "top_tier_envs": int((tiers >= 2).sum())}


def _size(default_envs, default_steps=400):
    return int(os.environ.get("PCC_FULL_SIZE_ENVS", default_envs)), int(os.environ.get("PCC_FULL_SIZE_STEPS", default_steps))


def _not_silently_reduced(n, steps, full_envs, full_steps=400):
    """PCC_FULL_SIZE_ENVS / PCC_FULL_SIZE_STEPS shrink these tests (tests/test_variants.py runs a reduced config 3 through the
    one-launch step).  A reduced run must not read as a full-size pass: after its comparison HAS passed it reports itself as
    skipped, with the size it ran at, in the summary."""
    if n < full_envs or steps < full_steps:
        pytest.skip("REDUCED run: %d envs x %d steps matched the oracle (full size is %d x %d: not a full-size result)"
                    % (n, steps, full_envs, full_steps))


def test_config3_every_env_every_step_matches_the_oracle():
    """BASELINE.json configs[2] as bench.py runs it: 65 536 envs, randomized links, one sender, 400-step episode, then
    the auto-reset and 20 steps of the second episode."""
    n, steps = _size(65536)
    info = _run_and_compare(n, 1, seed=0, n_steps=steps, extra_steps=20)
    if n >= 65536 and steps >= 400:
        # the point of the size: the largest envs of a whole episode are in the comparison
        assert info["max_packets_in_a_step"] >= 4096, info   # (the team items' threshold: the largest envs went that way)
    print("config 3 whole batch:", info)
    _not_silently_reduced(n, steps, 65536)


def test_config5_every_env_every_step_matches_the_oracle():
    """BASELINE.json configs[4]: 32 768 envs x 2 senders on one bottleneck, whole episode + 10 steps of the next."""
    n, steps = _size(32768)
    n = min(n, 32768)
    info = _run_and_compare(n, 2, seed=4, n_steps=steps, extra_steps=10)
    print("config 5 whole batch:", info)
    _not_silently_reduced(n, steps, 32768)


@pytest.mark.parametrize("n_senders", [1, 2])
def test_staggered_phases_every_env_every_step_matches_the_oracle(n_senders):
    """bench.py --stagger's schedule at 8 192 envs (4 096 with two senders): env i is reset (masked) at step i % P of a
    pre-roll, so the episode phases are spread uniformly and from then on some envs finish in EVERY step -- the path where
    the host cannot know who finishes: the retire half resets a finished env and files it as a restart item, the next send
    launch runs its warm-up intervals.  Every env, every step from the first masked reset on, against the oracle: with
    Philox uniforms an episode depends on (env id, episode index, actions) only, so env i's k-th episode is one oracle
    call with first_episode=k."""
    n = int(os.environ.get("PCC_STAGGER_ENVS", 8192 if n_senders == 1 else 4096))
    S, P = n_senders, 100                      # episode length P: three episode generations in 260 steps
    T = 260
    seed = 9
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=seed, n_senders=S, record_steps=True, auto_reset=True, max_steps=P)
    env.reset()
    gen = torch.Generator(device=DEV).manual_seed(77)
    acts = torch.rand((T, n, S), generator=gen, device=DEV, dtype=torch.float32) * 2.5 - 1   # U(-1, 1.5): rates climb, rings promote
    phase = torch.arange(n, device=DEV) % P
    rows = torch.empty((T, n, S, 19), dtype=torch.float64, device=DEV)
    obs = torch.empty((T, n, S, env.obs_dim), dtype=torch.float32, device=DEV)
    for t in range(T):
        if 0 < t < P:
            env.reset(phase == t)              # (phase 0 keeps the episode the full reset started)
        o, r, d, info = env.step(acts[t])
        rows[t] = info["steps"].reshape(n, S, 19)
        obs[t] = o.reshape(n, S, -1)
    torch.cuda.synchronize()
    env.check_flags()
    # env i: episode 0 = steps [0, ph) (cut short by its masked reset; ph = 0: a whole episode), then episodes start at
    # ph, ph + P, ph + 2P, ... (ph = 0: P, 2P, ...); episode index k starts at step start_k
    a_host = acts.to(torch.float64).permute(1, 0, 2).contiguous().cpu().numpy()
    ph = (np.arange(n) % P)
    got_rows = rows.permute(1, 2, 0, 3).contiguous().cpu().numpy()    # [n, S, T, 19]
    got_obs = obs.permute(1, 2, 0, 3).contiguous().cpu().numpy()
    problems = []
    for k in range(0, 4):
        # start step of episode k per env
        start = np.where(ph == 0, k * P, (0 if k == 0 else ph + (k - 1) * P))
        length = np.where((k == 0) & (ph > 0), ph, P)
        length = np.minimum(length, T - start)
        live = length > 0
        if not live.any():
            break
        a = np.zeros((n, P, S))
        for i in np.nonzero(live)[0]:
            a[i, :length[i]] = a_host[i, start[i]:start[i] + length[i]]
        ref = oracle.run_batch(a, n_senders=S, rng_mode=oracle.RNG_PHILOX, seed=seed, first_episode=k)
        ref_steps = ref["steps"].reshape(n, S, P, 19)
        ref_obs = ref["obs"].reshape(n, S, P, -1).astype(np.float32)
        ref_obs0 = ref["obs0"].reshape(n, S, -1).astype(np.float32)
        for i in np.nonzero(live)[0]:
            s0, L = int(start[i]), int(length[i])
            if not np.array_equal(got_rows[i, :, s0:s0 + L], ref_steps[i, :, :L]):
                problems.append(_first_mismatch(got_rows[i:i + 1, :, s0:s0 + L], ref_steps[i:i + 1, :, :L],
                                                "env %d, episode %d (steps %d..%d)" % (i, k, s0, s0 + L - 1)))
            # observations: all but a finished episode's last row, which shows the next episode's first observation
            full = L == P
            Lo = L - 1 if full else L
            if not np.array_equal(got_obs[i, :, s0:s0 + Lo], ref_obs[i, :, :Lo]):
                problems.append(_first_mismatch(got_obs[i:i + 1, :, s0:s0 + Lo], ref_obs[i:i + 1, :, :Lo],
                                                "env %d, episode %d observations" % (i, k)))
            if k > 0 and s0 > 0 and (ph[i] == 0 or k > 1):
                # this episode began with an auto-reset inside step s0 - 1: that step's observation row is its obs0
                if not np.array_equal(got_obs[i, :, s0 - 1], ref_obs0[i]):
                    problems.append("env %d: observation after the auto-reset into episode %d differs" % (i, k))
            if len(problems) > 8:
                break
        assert not problems, "\n".join(p for p in problems if p)
    # the envs' own episode ends (from step P on) went through the shadows: next episodes prepared ahead of time and swapped in
    stats = env.restart_stats()
    assert stats["shadow_swaps"] > n and stats["shadow_swaps"] > 20 * stats["restart_list"], stats
    env.close()


def test_saturating_policy_at_full_size_with_default_pools():
    """65 536 envs driven towards the rate limit (U(0, 2) actions, like the reference-generated saturating_0_2 goldens) for a
    whole episode with the DEFAULT ring pools: no env may be flagged -- a policy that learns to fill its links must not end a
    training run with PCC_FLAG_POOL_EXHAUSTED (round 3's pools were sized for U(-1, 1) policies and did) -- and the first 512
    envs are compared with the oracle, every step.  ns:235-241, 275-281 (the rate climbs to MAX_RATE and stays there)."""
    n, steps = _size(65536)
    M = min(512, n)
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=2, record_steps=True, auto_reset=False)
    env.reset()
    gen = torch.Generator(device=DEV).manual_seed(99)
    rows, acts = [], []
    for t in range(steps):
        a = torch.rand((n,), generator=gen, device=DEV, dtype=torch.float32) * 2
        o, r, d, info = env.step(a)
        rows.append(info["steps"][:M].clone())
        acts.append(a[:M].clone())
    torch.cuda.synchronize()
    flags = env.state("flags").cpu().numpy()
    assert not flags.any(), "flagged envs: %d (pool exhausted: %d, ring overflow: %d)" % (
        int((flags != 0).sum()), int(((flags & pcc_rl_amd.native.PCC_FLAG_POOL_EXHAUSTED) != 0).sum()),
        int(((flags & pcc_rl_amd.native.PCC_FLAG_RING_OVERFLOW) != 0).sum()))
    tiers = env.state("ring_tier").cpu().numpy()
    ref = oracle.run_batch(torch.stack(acts, 1).to(torch.float64).cpu().numpy(), rng_mode=oracle.RNG_PHILOX, seed=2, want_obs=False)
    problem = _first_mismatch(torch.stack(rows, 1).cpu().numpy(), ref["steps"], "saturating policy, first %d envs" % M)
    assert not problem, problem
    print("saturating policy: senders by ring tier", [int((tiers == c).sum()) for c in range(4)], "device GB", env.device_bytes / 1e9)
    env.close()


def test_a_partition_that_outgrows_its_share_of_a_pool_takes_the_others():
    """The ring pools are cut into one stack per partition of the batch (pcc_dev.h "partitions": an XCD keeps to one eighth of
    every pool).  Here only the envs of partition 0 are driven to the rate limit, with pools of a quarter of the senders: they
    need four times their own share of tiers 1 and 2 and take it from the other partitions' stacks -- no flag, slots of
    foreign shares in use, results equal to the oracle -- and when the pools are made too small for them the exhaustion is
    flagged, never silent."""
    n, steps, M = 8192, 160, 192
    native = pcc_rl_amd.native
    for divisors, expect_flag in (((4, 4, 4), False), ((64, 64, 64), True)):
        env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=5, record_steps=True, auto_reset=False, ring_pools=divisors)
        env.reset()
        part0 = torch.arange(n, device=DEV) < n // 8
        gen = torch.Generator(device=DEV).manual_seed(7)
        rows, acts = [], []
        for t in range(steps):
            up = torch.rand((n,), generator=gen, device=DEV, dtype=torch.float32) * 2          # U(0, 2): towards MAX_RATE
            a = torch.where(part0, up, torch.full_like(up, -1.0))                              # everybody else towards MIN_RATE
            o, r, d, info = env.step(a)
            rows.append(info["steps"][:M].clone())
            acts.append(a[:M].clone())
        torch.cuda.synchronize()
        flags = env.state("flags").cpu().numpy()
        tiers = env.state("ring_tier").cpu().numpy()
        if expect_flag:
            assert ((flags & native.PCC_FLAG_POOL_EXHAUSTED) != 0).any(), "pools of 256 slots per tier cannot hold 1 024 saturated senders"
        else:
            assert not flags.any(), "flagged envs: %d" % int((flags != 0).sum())
            promoted = int((tiers[: n // 8] >= 1).sum())
            assert promoted > n // 8 // 4 * 2, "partition 0 holds %d pool slots: not beyond its own share of %d" % (promoted, n // 8 // 4)
            ref = oracle.run_batch(torch.stack(acts, 1).to(torch.float64).cpu().numpy(), rng_mode=oracle.RNG_PHILOX, seed=5, want_obs=False)
            problem = _first_mismatch(torch.stack(rows, 1).cpu().numpy(), ref["steps"], "partition 0 on borrowed pool slots, first %d envs" % M)
            assert not problem, problem
        env.close()
