"""GPU parity tests: the HIP path, called through the C ABI (ctypes -> libpcc_sim.so), against
(1) the golden vectors generated from the reference and (2) the CPU oracle on seeded inputs.
Integer counts must be equal; every float (clock, run_dur, reward, all 12 metrics) is
compared bit-for-bit; float32 observations must equal the float32 cast of the oracle's."""
import os

import numpy as np
import pytest
import torch

import oracle
import pcc_rl_amd

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


def load(name):
    with np.load(os.path.join(G, name + ".npz")) as z:
        return {k: z[k] for k in z.files}


def run_gpu(env, actions, n_steps):
    """actions [N, T(, S)] float64 numpy; returns steps [N, (S,) T, 19], obs [N, (S,) T, HF]."""
    acts = torch.as_tensor(actions, dtype=torch.float64, device=DEV)
    rows, obs, dones = [], [], []
    for t in range(n_steps):
        o, r, d, info = env.step(acts[:, t])
        rows.append(info["steps"].clone())
        obs.append(o.clone())
        dones.append(d.clone())
    torch.cuda.synchronize()
    env.check_flags()
    ax = 1 if env.n_senders == 1 else 2
    return (torch.stack(rows, ax).cpu().numpy(), torch.stack(obs, ax).cpu().numpy(),
            torch.stack(dones, 1).cpu().numpy())


def golden_env(d, history_len=10, features=None, n_senders=1):
    """Batch of the golden cases in trace mode: parameters from the fixture, loss uniforms =
    the MT19937 stream random.Random(seed) continues with after the parameter draws."""
    n = d["seed"].shape[0]
    feats = [str(f) for f in d["features"]] if features is None and "features" in d else (features or pcc_rl_amd.DEFAULT_FEATURES)
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, history_len=history_len, features=feats,
                                       n_senders=n_senders, record_steps=True, auto_reset=False)
    p = d["params"]
    if n_senders == 1:
        env.set_link_params(p[:, 0], p[:, 1], np.round(p[:, 2]), p[:, 3], p[:, 4])
    else:
        env.set_link_params(p[:, 0], p[:, 1], np.round(p[:, 2]), p[:, 3], p[:, 4:6])
    k = int((d["rng"][:, 1] - d["rng"][:, 0]).max())
    trace = np.stack([oracle.mt_uniforms(int(s), k, skip=int(o)) for s, o in zip(d["seed"], d["rng"][:, 0])])
    env.set_loss_trace(trace)
    return env


@pytest.mark.parametrize("name", ["default_pm1", "saturating_0_2", "clamp_pm30", "allfeat_h3", "fixed_cfg2",
                                  "fixed_q1", "fixed_lossy", "fixed_deepq"])
def test_golden_vectors_bit_exact(name):
    d = load(name)
    env = golden_env(d, history_len=int(d["history_len"]))
    obs0 = env.reset().cpu().numpy()
    assert np.array_equal(obs0, d["obs0"].astype(np.float32))
    assert np.array_equal(env.state("now").cpu().numpy(), d["warm"][:, 0])
    in_flight = (env.state("acc_tail") - env.state("acc_head") + env.state("drop_tail") - env.state("drop_head"))[0].cpu().numpy()
    # heap length after warm-up = packets in flight + the pending SEND
    assert np.array_equal(in_flight + 1, d["warm"][:, 1].astype(np.int64))
    T = d["actions"].shape[1]
    steps, obs, done = run_gpu(env, d["actions"], T)
    assert np.array_equal(steps[..., :3], d["steps"][..., :3]), "sent/acked/lost"
    assert np.array_equal(steps, d["steps"]), "clocks, reward, metrics"
    nf = d["obs_tail"].shape[2]
    assert np.array_equal(obs[..., -nf:], d["obs_tail"].astype(np.float32))
    k = d["obs_full"].shape[0]
    assert np.array_equal(obs[:k], d["obs_full"].astype(np.float32))
    assert np.array_equal(done, d["done"])
    env.close()


def test_two_consecutive_episodes_of_the_reference_on_one_handle():
    """tests/golden/two_episodes.npz: the reference env object ran two episodes back to back (rows i, i + 1 = episodes 0, 1
    of one env: new links from the same random stream).  Here one handle runs episode 0, is reset with the second
    episode's links and the stream position the reference had reached, and must reproduce episode 1 bit for bit --
    rings, tiers, cursors and the connection minimum all start over."""
    d = load("two_episodes")
    first = {k: (v[0::2] if getattr(v, "ndim", 0) and v.shape[0] == d["seed"].shape[0] else v) for k, v in d.items()}
    second = {k: (v[1::2] if getattr(v, "ndim", 0) and v.shape[0] == d["seed"].shape[0] else v) for k, v in d.items()}
    assert (second["episode"] == 1).all() and (first["seed"] == second["seed"]).all()
    env = golden_env(first, history_len=int(d["history_len"]))
    for ep, dd in enumerate((first, second)):
        if ep:
            p = dd["params"]
            env.set_link_params(p[:, 0], p[:, 1], np.round(p[:, 2]), p[:, 3], p[:, 4])
            k = int((dd["rng"][:, 1] - dd["rng"][:, 0]).max())
            env.set_loss_trace(np.stack([oracle.mt_uniforms(int(sd), k, skip=int(o)) for sd, o in zip(dd["seed"], dd["rng"][:, 0])]))
        obs0 = env.reset().cpu().numpy()
        assert np.array_equal(obs0, dd["obs0"].astype(np.float32))
        assert np.array_equal(env.state("now").cpu().numpy(), dd["warm"][:, 0])
        T = dd["actions"].shape[1]
        steps, obs, done = run_gpu(env, dd["actions"], T)
        assert np.array_equal(steps, dd["steps"]), "episode %d" % ep
        assert np.array_equal(obs[..., -3:], dd["obs_tail"].astype(np.float32))
        assert np.array_equal(done, dd["done"])
    env.close()


def test_two_sender_golden_bit_exact():
    d = load("two_sender")
    d["features"] = np.array(pcc_rl_amd.DEFAULT_FEATURES.split(","))
    env = golden_env(d, n_senders=2)
    env.reset()
    assert np.array_equal(env.state("now").cpu().numpy(), d["warm"][:, 0])
    T = d["actions"].shape[1]
    steps, obs, _ = run_gpu(env, d["actions"], T)
    assert np.array_equal(steps[..., :3], d["steps"][..., :3])
    assert np.array_equal(steps, d["steps"])
    assert np.array_equal(obs[..., -3:], d["obs_tail"].astype(np.float32))
    env.close()


def test_two_sender_golden_with_another_observation_shape_bit_exact():
    """tests/golden/two_sender_allfeat_h3.npz (the reference's engine with two senders, all 12 features, three intervals of
    history): every column and the WHOLE observation of both senders at every step -- the history / observation writer with
    S = 2, F != 3, H != 10."""
    d = load("two_sender_allfeat_h3")
    env = golden_env(d, history_len=int(d["history_len"]), n_senders=2)
    env.reset()
    assert np.array_equal(env.state("now").cpu().numpy(), d["warm"][:, 0])
    T = d["actions"].shape[1]
    steps, obs, _ = run_gpu(env, d["actions"], T)
    assert np.array_equal(steps[..., :3], d["steps"][..., :3])
    assert np.array_equal(steps, d["steps"])
    assert obs.shape == d["obs_full"].shape, (obs.shape, d["obs_full"].shape)
    assert np.array_equal(obs, d["obs_full"].astype(np.float32))
    env.close()


@pytest.mark.parametrize("n_envs,n_steps,seed,n_senders,history_len", [(8193, 40, 21, 1, 10), (600, 60, 22, 2, 1), (8193, 30, 23, 2, 4)])
def test_philox_batches_with_odd_shapes_match_oracle(n_envs, n_steps, seed, n_senders, history_len):
    """Partitions with a one-env tail (8 193 envs = 8 partitions of 1 088 envs, the last one holding 577), two senders with a
    history of one interval, two senders over ragged partitions: device-drawn parameters against the oracle, every column and
    the whole observation."""
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, n_senders=n_senders, history_len=history_len,
                                       record_steps=True, auto_reset=False)
    obs0 = env.reset().cpu().numpy()
    rs = np.random.RandomState(seed)
    acts = rs.uniform(-1, 1.5, (n_envs, n_steps, n_senders) if n_senders > 1 else (n_envs, n_steps))
    steps, obs, done = run_gpu(env, acts, n_steps)
    ref = oracle.run_batch(acts, n_senders=n_senders, history_len=history_len, rng_mode=oracle.RNG_PHILOX, seed=seed)
    assert np.array_equal(obs0, ref["obs0"].astype(np.float32))
    bad = np.argwhere((steps[..., :3] != ref["steps"][..., :3]).any(axis=tuple(range(1, steps.ndim))))
    assert bad.size == 0, "envs with count mismatches: %s" % bad[:10].ravel()
    assert np.array_equal(steps, ref["steps"])
    assert np.array_equal(obs, ref["obs"].astype(np.float32))
    env.check_flags()
    env.close()


@pytest.mark.parametrize("n_envs,n_steps,seed", [(4096, 60, 11), (777, 400, 5)])
def test_philox_batches_match_oracle(n_envs, n_steps, seed):
    """Randomized parameters drawn on the device vs the oracle drawing them on the host from the
    same Philox stream; N not a multiple of the wavefront exercises the ragged tail."""
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, env_gid_base=1000, record_steps=True,
                                       auto_reset=False)
    obs0 = env.reset().cpu().numpy()
    rs = np.random.RandomState(seed)
    acts = rs.uniform(-1, 1, (n_envs, n_steps))
    steps, obs, done = run_gpu(env, acts, n_steps)
    ref = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=seed, env_gid_base=1000)
    # parameters (bw, dl, queue, lr, rate0)
    assert np.array_equal(env.state("bw").cpu().numpy(), ref["params"][:, 0])
    assert np.array_equal(env.state("dl").cpu().numpy(), ref["params"][:, 1])
    assert np.array_equal(env.state("lr").cpu().numpy(), ref["params"][:, 3])
    assert np.array_equal(env.state("rate0")[0].cpu().numpy(), ref["params"][:, 4])
    assert np.array_equal(obs0, ref["obs0"].astype(np.float32))
    bad = np.argwhere((steps[..., :3] != ref["steps"][..., :3]).any(axis=(1, 2)))
    assert bad.size == 0, "envs with count mismatches: %s" % bad[:10].ravel()
    assert np.array_equal(steps, ref["steps"])
    assert np.array_equal(obs, ref["obs"].astype(np.float32))
    assert done[:, -1].all() == (n_steps >= 400)
    env.close()


def test_config2_fixed_params_all_envs_identical():
    """BASELINE.json configs[1]: 4096 envs, fixed link (200 pkt/s, 30 ms, queue 5, no loss),
    rate0 60: deterministic, so every env must reproduce the golden episode exactly."""
    d = load("fixed_cfg2")
    N = 4096
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=DEV, record_steps=True, auto_reset=False,
                                       link_params=(200.0, 0.03, 5.0, 0.0, 60.0))
    env.reset()
    acts = np.tile(d["actions"][0][None, :], (N, 1))
    steps, obs, _ = run_gpu(env, acts, 400)
    assert np.array_equal(steps[0], d["steps"][0])
    assert (steps == steps[0:1]).all()
    env.close()


def test_auto_reset_and_second_episode_match_oracle():
    n_envs, seed = 256, 3
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, record_steps=True, auto_reset=True,
                                       max_steps=25)
    env.reset()
    rs = np.random.RandomState(1)
    acts = rs.uniform(-1, 1, (n_envs, 50))
    steps, obs, done = run_gpu(env, acts, 50)
    assert done[:, 24].all() and done[:, 49].all() and done.sum() == 2 * n_envs
    # oracle: same env objects run two 25-step episodes (MAX_STEPS is 400 there, so drive resets by hand)
    ref1 = oracle.run_batch(acts[:, :25], rng_mode=oracle.RNG_PHILOX, seed=seed)
    assert np.array_equal(steps[:, :25], ref1["steps"])
    # after the auto-reset the observation row is the fresh-episode observation
    assert np.array_equal(obs[:, 24], ref1["obs0"].astype(np.float32))
    # episode index 1 of the same envs (run_batch returns the last episode; with Philox uniforms an
    # episode depends on its index and its actions only): every column and the observations
    ref2 = oracle.run_batch(acts[:, 25:], rng_mode=oracle.RNG_PHILOX, seed=seed, n_episodes=2)
    assert np.array_equal(steps[:, 25:, :3], ref2["steps"][..., :3])
    assert np.array_equal(steps[:, 25:], ref2["steps"])
    assert np.array_equal(obs[:, 25:49], ref2["obs"][:, :24].astype(np.float32))
    ret = env.episode_returns().cpu().numpy()
    assert np.allclose(ret, ref2["steps"][..., 6].sum(1), rtol=1e-12)
    env.close()


@pytest.mark.parametrize("knobs", [dict(), dict(heavy_predict=128.0, takeover_lanes=4), dict(team_predict=300.0)])
def test_conservation_and_queue_bounds_at_full_size(knobs):
    """Size-independent properties at BASELINE's 65 536 envs: every packet sent is acked, lost
    or still in flight; the queue never exceeds its limit; clocks only move forward.  And the
    first 512 envs of the full batch -- wherever the work lists put them -- against the oracle,
    bit for bit."""
    N = 65536
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=DEV, seed=0, record_steps=True, auto_reset=False)
    env.set_tuning(**knobs)
    env.reset()
    gen = torch.Generator(device=DEV).manual_seed(0)
    sent0 = (env.state("acc_tail") + env.state("drop_tail"))[0].clone().long()
    head0 = (env.state("acc_head") + env.state("drop_head"))[0].clone().long()
    acked = torch.zeros(N, dtype=torch.float64, device=DEV)
    sent = torch.zeros_like(acked)
    now_prev = env.state("now").clone()
    M, rows_m, acts_m = 512, [], []
    for t in range(40):
        a = torch.rand((N,), generator=gen, device=DEV, dtype=torch.float64) * 2 - 1
        o, r, d, info = env.step(a)
        s = info["steps"]
        rows_m.append(s[:M].clone())
        acts_m.append(a[:M].clone())
        sent += s[:, 0]
        acked += s[:, 1] + s[:, 2]
        now = env.state("now")
        assert bool((now > now_prev).all())
        now_prev = now.clone()
        assert bool((env.state("queue_delay") <= env.state("maxq")).all())
        assert bool(torch.isfinite(o).all()) and bool(torch.isfinite(r).all())
    env.check_flags()
    tail = (env.state("acc_tail") + env.state("drop_tail"))[0].long()
    head = (env.state("acc_head") + env.state("drop_head"))[0].long()
    assert bool(((tail - sent0).double() == sent).all())
    assert bool(((head - head0).double() == acked).all())
    assert bool((env.state("acc_tail") >= env.state("acc_head")).all())
    assert bool((env.state("drop_tail") >= env.state("drop_head")).all())
    ref = oracle.run_batch(torch.stack(acts_m, 1).cpu().numpy(), rng_mode=oracle.RNG_PHILOX, seed=0, want_obs=False)
    assert np.array_equal(torch.stack(rows_m, 1).cpu().numpy(), ref["steps"])
    env.close()


def test_old_gym_adapter_drop_in():
    """An agent loop written against the reference env (reset / step([a]) / done) runs unchanged."""
    d = load("default_pm1")
    p = d["params"][0]
    env = pcc_rl_amd.SimulatedNetworkEnv(device=DEV, link_params=(p[0], p[1], round(p[2]), p[3], p[4]))
    with pytest.raises(TypeError):
        env.step([0.0])
    assert env.seed(5) == [5]
    env._env.set_loss_trace(oracle.mt_uniforms(0, int(d["rng"][0, 1]), skip=10)[None, :])
    obs = env.reset()
    assert obs.shape == (30,) and obs.dtype == np.float64   # float64 at run time, like the reference's
    assert env.observation_space.shape == (30,) and env.action_space.shape == (1,)
    assert np.array_equal(obs, np.tile([0.0, 1.0, 1.0], 10))
    total, done, t = 0.0, False, 0
    while not done:
        obs, rew, done, info = env.step([d["actions"][0, t]])
        assert isinstance(rew, float) and isinstance(done, bool) and info == {}
        assert rew == d["steps"][0, t, 6]
        assert obs.dtype == np.float64 and np.array_equal(obs[-3:], d["obs_tail"][0, t])   # the reference's own float64 values
        assert np.array_equal(obs, d["obs_full"][0, t])   # ... the whole 30-vector, every step
        total += rew
        t += 1
    assert t == 400 and total == 625.6196947931776   # SURVEY.md section 8(c) KAT
    assert len(env.event_record["Events"]) == 400
    # dump_events_to_file (ns:493-496): the JSON the reference writes -- {"Events": [one dict per step]}, its nine fields in
    # its order (ns:422-436), the numbers of the golden step record
    import json, tempfile
    from pcc_rl_amd import native
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "pcc_env_log_run_0.json")
        env.dump_events_to_file(path)
        with open(path) as f:
            log = json.load(f)
    assert list(log.keys()) == ["Events"] and len(log["Events"]) == 400
    col = native.STEP_COLUMNS.index
    fields = [("Reward", "reward"), ("Send Rate", "send rate"), ("Throughput", "recv rate"), ("Latency", "avg latency"),
              ("Loss Rate", "loss ratio"), ("Latency Inflation", "sent latency inflation"), ("Latency Ratio", "latency ratio"),
              ("Send Ratio", "send ratio")]
    for k, ev in enumerate(log["Events"]):
        assert list(ev.keys()) == ["Name", "Time"] + [name for name, _ in fields]
        assert ev["Name"] == "Step" and ev["Time"] == k + 1
        for name, column in fields:
            assert ev[name] == d["steps"][0, k, col(column)], (k, name)   # (json round-trips a float64 exactly)
    env.close()


@pytest.mark.parametrize("knobs", [
    dict(takeover_lanes=64, round_packets=8),            # everything through the wave path after 8 packets
    dict(takeover_lanes=0, heavy_predict=1e18),          # lane-serial only
    dict(heavy_predict=0.0),                             # every env sent by the heavy wavefront from the start
    dict(send_envs_per_wave=7, round_packets=64, takeover_lanes=3),
    dict(takeover_lanes=0, heavy_predict=1e18),                  # lane rounds only
    dict(send_waves=1, heavy_predict=64.0),                      # few persistent wavefronts, many heavy items each
    dict(send_waves=32, heavy_predict=64.0),                     # more wavefronts than items
    dict(send_envs_per_wave=64, heavy_predict=256.0, takeover_lanes=8),
    dict(heavy_predict=16.0, round_packets=16),                  # nearly everything by wave passes of every regime
    dict(heavy_predict=0.0, team_predict=0.0),                   # every env a TEAM item: four wavefronts, 1 024 positions per pass
    dict(heavy_predict=16.0, team_predict=100.0, round_packets=16),   # lane rounds, wave passes and team passes side by side
    dict(team_predict=1e18),                                     # no team items: the giants on one wavefront
    dict(heavy_predict=64.0, team_predict=64.0, send_waves=1),   # more team items than team workgroups
    dict(retire_wide_predict=0.0),                               # retire half: every env by 16 lanes
    dict(retire_wide_predict=1e18),                              # ... every env by 8 lanes (the giants' three sums in a row)
    dict(retire_wide_predict=40.0, heavy_predict=64.0),          # ... both kinds of workgroup, many of each
    dict(heavy_predict=32.0, heavy_item_packets=0.0),            # one env per wave-path item
    dict(heavy_predict=32.0, heavy_item_packets=1e6, team_predict=1e18),   # eight envs per wave-path item
])
def test_send_paths_are_exact_whatever_the_tuning(knobs):
    """The tuning knobs only choose WHICH exact send path runs (lane-serial rounds, the wave-wide
    serial pass, the accept-to-accept pass): every setting must reproduce the oracle bit for bit."""
    n_envs, n_steps, seed = 600, 120, 21
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, record_steps=True, auto_reset=False)
    env.set_tuning(**knobs)
    env.reset()
    rs = np.random.RandomState(seed)
    acts = rs.uniform(-1, 2, (n_envs, n_steps))   # upward drift: overloaded queues, many drops
    steps, obs, done = run_gpu(env, acts, n_steps)
    ref = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=seed)
    assert np.array_equal(steps[..., :3], ref["steps"][..., :3])
    assert np.array_equal(steps, ref["steps"])
    assert np.array_equal(obs, ref["obs"].astype(np.float32))
    env.close()


@pytest.mark.parametrize("knobs", [
    dict(),                                              # defaults: rounds, then the two-sender wave path
    dict(takeover_lanes=0),                              # lane-serial only
    dict(takeover_lanes=64, round_packets=8),            # merge-path wave passes for almost everything
    dict(takeover_lanes=64, round_packets=4, heavy_predict=200.0),
    dict(heavy_predict=100.0, send_waves=2),
    dict(retire_wide_predict=0.0),
    dict(retire_wide_predict=1e18),
])
def test_two_sender_philox_batches_match_oracle(knobs):
    """BASELINE.json configs[4] shape (two senders on one link) at a size the oracle finishes in
    seconds: one Philox stream per env consumed in merged send order, both senders' rows and observations."""
    n_envs, n_steps, seed = 500, 100, 33
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, n_senders=2, record_steps=True,
                                       auto_reset=False)
    env.set_tuning(**knobs)
    env.reset()
    rs = np.random.RandomState(seed)
    acts = rs.uniform(-1, 1.5, (n_envs, n_steps, 2))
    steps, obs, done = run_gpu(env, acts, n_steps)
    ref = oracle.run_batch(acts, n_senders=2, rng_mode=oracle.RNG_PHILOX, seed=seed)
    bad = np.argwhere((steps[..., :3] != ref["steps"][..., :3]).any(axis=tuple(range(1, steps.ndim))))
    assert bad.size == 0, "envs with count mismatches: %s" % bad[:10].ravel()
    assert np.array_equal(steps, ref["steps"])
    assert np.array_equal(obs, ref["obs"].astype(np.float32))
    env.close()


def test_two_sender_wave_path_across_powers_of_two_in_time_matches_oracle():
    """160 intervals of overdriven two-sender envs sent by the wave path alone: every env's send times cross 1, 2, 4, 8 ...
    seconds inside some interval.  A pass of heavy_mi2 ends at the top of the binade of the send times (t0 + k G is exact
    only below it) and the next send time is ns:161 on the last packet's time -- round 6; before, such an interval fell back
    to 62 packets per pass."""
    n_envs, n_steps, seed = 256, 160, 47
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, n_senders=2, record_steps=True, auto_reset=False)
    env.set_tuning(takeover_lanes=64, round_packets=4, heavy_predict=100.0)
    env.reset()
    rs = np.random.RandomState(seed)
    acts = rs.uniform(-0.6, 1.6, (n_envs, n_steps, 2))
    steps, obs, done = run_gpu(env, acts, n_steps)
    ref = oracle.run_batch(acts, n_senders=2, rng_mode=oracle.RNG_PHILOX, seed=seed)
    assert np.array_equal(steps, ref["steps"])
    assert np.array_equal(obs, ref["obs"].astype(np.float32))
    env.close()


def test_two_sender_wave_path_on_golden_trace():
    d = load("two_sender")
    d["features"] = np.array(pcc_rl_amd.DEFAULT_FEATURES.split(","))
    env = golden_env(d, n_senders=2)
    env.set_tuning(takeover_lanes=64, round_packets=4)
    env.reset()
    env.set_tuning(takeover_lanes=64, round_packets=4)
    T = d["actions"].shape[1]
    steps, obs, _ = run_gpu(env, d["actions"], T)
    assert np.array_equal(steps, d["steps"])
    env.close()


def test_wave_path_on_golden_traces():
    """Trace mode through the wave path: the saturating and deep-queue goldens with every env handed
    to the heavy wavefront."""
    for name in ("saturating_0_2", "fixed_deepq", "fixed_lossy"):
        d = load(name)
        env = golden_env(d, history_len=int(d["history_len"]))
        env.set_tuning(heavy_predict=0.0)
        env.reset()
        env.set_tuning(heavy_predict=0.0, takeover_lanes=64, round_packets=4)
        steps, obs, done = run_gpu(env, d["actions"], d["actions"].shape[1])
        assert np.array_equal(steps, d["steps"]), name
        env.close()


def test_team_path_on_golden_traces():
    """Trace mode through the team path (a whole workgroup per env, heavy_mi<.., 4>): the reference's own episodes with
    every env a team item from its first interval on."""
    for name in ("saturating_0_2", "fixed_deepq", "fixed_lossy", "default_pm1"):
        d = load(name)
        env = golden_env(d, history_len=int(d["history_len"]))
        env.reset()
        env.set_tuning(heavy_predict=0.0, team_predict=0.0)
        steps, obs, done = run_gpu(env, d["actions"], d["actions"].shape[1])
        assert np.array_equal(steps, d["steps"]), name
        env.close()


def test_step_halves_and_protocol_errors():
    env = pcc_rl_amd.BatchedNetworkEnv(64, device=DEV, seed=1)
    with pytest.raises(pcc_rl_amd.PccError):
        env.step(torch.zeros(64, device=DEV))          # step before reset
    env.reset()
    a = torch.zeros(64, device=DEV)
    env.step_send(a)
    with pytest.raises(pcc_rl_amd.PccError):
        env.step_send(a)                               # two sends without a retire
    o, r, d, _ = env.step_retire()
    with pytest.raises(pcc_rl_amd.PccError):
        env.step_retire()                              # retire without a send
    env2 = pcc_rl_amd.BatchedNetworkEnv(64, device=DEV, seed=1)
    env2.reset()
    o2, r2, d2, _ = env2.step(a)
    assert torch.equal(o, o2) and torch.equal(r, r2)
    env.close(); env2.close()


def test_ring_overflow_is_flagged_not_silent():
    """A ring too small for the packets in flight sets the sticky flag (results are then invalid)."""
    env = pcc_rl_amd.BatchedNetworkEnv(64, device=DEV, seed=0, ring_capacity=16, auto_reset=False,
                                       link_params=(100.0, 0.4, 2000.0, 0.0, 900.0))
    env.reset()
    for _ in range(5):
        env.step(torch.ones(64, device=DEV))
    with pytest.raises(pcc_rl_amd.PccError):
        env.check_flags()
    env.close()


@pytest.mark.parametrize("params,feats,hist", [
    ((300.0, 0.05, 40.0, 1.0, 250.0), "loss ratio,send rate", 1),        # every packet lost: no acks, no RTTs
    ((120.0, 0.2, 1.0, 0.0, 500.0), "recv rate,avg latency,conn min latency,latency increase", 4),  # queue of one packet
    ((450.0, 0.012, 3000.0, 0.02, 44.0), "send dur,recv dur,ack latency inflation", 2),             # starts at MIN_RATE
])
def test_edge_links_match_oracle(params, feats, hist):
    n_envs, n_steps = 64, 80
    rs = np.random.RandomState(9)
    acts = rs.uniform(-1, 1, (n_envs, n_steps))
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=4, history_len=hist, features=feats,
                                       record_steps=True, auto_reset=False, link_params=params)
    obs0 = env.reset().cpu().numpy()
    steps, obs, done = run_gpu(env, acts, n_steps)
    p = np.tile(np.array(params)[None, :], (n_envs, 1))
    ref = oracle.run_batch(acts, history_len=hist, features=feats, rng_mode=oracle.RNG_PHILOX, seed=4, params=p)
    assert np.array_equal(obs0, ref["obs0"].astype(np.float32))
    assert np.array_equal(steps, ref["steps"])
    assert np.array_equal(obs, ref["obs"].astype(np.float32))
    env.close()


@pytest.mark.parametrize("params", [
    (100.0, 0.1, 3.0, 0.0, 1000.0),     # ten times overdriven: ~9 drops between two accepted packets
    (45.0, 0.08, 2.0, 0.01, 1000.0),    # twenty-two times: near groups longer than a 16-lane group
    (250.0, 0.05, 1.0, 0.03, 900.0),    # a queue of one packet, random losses in between
])
@pytest.mark.parametrize("wide", [0.0, 1e9])
def test_long_drop_runs_match_oracle(params, wide):
    """Overdriven links drop packets in runs, and consecutive drops arrive at mathematically equal times: the retire half
    orders such near groups exactly around its boundaries (fix_drop_boundary_g / drop_hop1_candidate_g: a group of 8 or 16
    lanes looks at a window of records at a time, the lead lane walks windows longer than the group).  Every env retired
    by 16 lanes (wide = 0) and every env by 8 (1e9); windows shorter and longer than either; ns:111, 141-146, 161, 178
    (the heap orders equal times by latency, then by the dropped flag)."""
    n_envs, n_steps = 192, 70
    rs = np.random.RandomState(21)
    acts = rs.uniform(-0.5, 1.5, (n_envs, n_steps))
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=8, record_steps=True, auto_reset=False, link_params=params)
    env.set_tuning(retire_wide_predict=wide, list_min_envs=0)
    obs0 = env.reset().cpu().numpy()
    steps, obs, done = run_gpu(env, acts, n_steps)
    p = np.tile(np.array(params)[None, :], (n_envs, 1))
    ref = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=8, params=p)
    assert np.array_equal(obs0, ref["obs0"].astype(np.float32))
    assert steps[..., 2].sum() > 5 * steps[..., 1].sum() or params[0] > 200, "the links are meant to lose most of their packets"
    assert np.array_equal(steps, ref["steps"])
    assert np.array_equal(obs, ref["obs"].astype(np.float32))
    env.close()


def test_masked_reset_only_touches_selected_envs():
    n = 128
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=2, record_steps=True, auto_reset=False)
    env.reset()
    a = torch.zeros(n, device=DEV)
    for _ in range(10):
        env.step(a)
    before_now = env.state("now").clone()
    before_steps = env.state("steps").clone()
    mask = torch.zeros(n, dtype=torch.bool, device=DEV)
    mask[::3] = True
    env.reset(mask)
    now, steps, ep = env.state("now"), env.state("steps"), env.state("episode")
    assert bool((steps[mask] == 0).all()) and bool((steps[~mask] == before_steps[~mask]).all())
    assert bool((now[~mask] == before_now[~mask]).all())
    assert bool((ep[mask] == 2).all()) and bool((ep[~mask] == 1).all())
    # both groups keep stepping consistently afterwards (second episode of the reset ones = oracle's)
    o, r, d, info = env.step(a)
    assert bool(torch.isfinite(o).all())
    env.check_flags()
    env.close()


def test_envs_out_of_lockstep_match_oracle():
    """Masked resets at staggered steps, then auto-resets at each env's own episode end (the path where
    the host cannot know which step finishes an episode: the retire half marks a finished env, the next send
    half gives it new links and runs its warm-up intervals as a restart item).
    Every env against its own oracle object driven through the same schedule -- all 19 columns and the
    observations, bit for bit, across episode boundaries."""
    n, seed, max_steps, T = 96, 11, 30, 85
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=seed, record_steps=True, auto_reset=True, max_steps=max_steps)
    oenvs = []
    for i in range(n):
        o = oracle.OracleEnv()
        o.rng_philox(seed, i)
        oenvs.append(o)
    obs = env.reset().cpu().numpy()
    oobs = np.stack([o.reset() for o in oenvs])
    assert np.array_equal(obs, oobs.astype(np.float32))
    osteps = np.zeros(n, dtype=int)
    rs = np.random.RandomState(5)
    idx = np.arange(n)
    for t in range(T):
        if t % 7 == 3:
            mask = (idx % 5) == ((t // 7) % 5)
            got = env.reset(torch.as_tensor(mask)).cpu().numpy()
            for i in idx[mask]:
                want = oenvs[i].reset()
                osteps[i] = 0
                assert np.array_equal(got[i], want.astype(np.float32)), (t, i)
        a = rs.uniform(-1, 1.5, n)
        o_gpu, r_gpu, d_gpu, info = env.step(torch.as_tensor(a, device=DEV))
        rows = info["steps"].cpu().numpy()
        o_gpu, d_gpu = o_gpu.cpu().numpy(), d_gpu.cpu().numpy()
        for i in range(n):
            o_ref, r_ref, _, _ = oenvs[i].step(a[i])
            osteps[i] += 1
            done = osteps[i] >= max_steps
            assert np.array_equal(rows[i], oenvs[i].last_row[0]), (t, i)
            assert bool(d_gpu[i]) == done, (t, i)
            if done:       # auto-reset: the row returned is the first observation of the next episode
                o_ref = oenvs[i].reset()
                osteps[i] = 0
            assert np.array_equal(o_gpu[i], o_ref.astype(np.float32)), (t, i)
        if t % 9 == 4 or t in (29, 30):
            # reading the state in between (t = 29: right after the envs that were never masked finished) shows every
            # env after its reset and warm-up intervals, whether or not the library has run them yet
            assert np.array_equal(env.state("steps").cpu().numpy(), osteps), t
            assert np.array_equal(env.state("now").cpu().numpy(), np.array([o.cur_time for o in oenvs])), t
    env.check_flags()
    env.close()


def test_new_link_params_out_of_lockstep_apply_at_every_envs_next_reset():
    """set_link_params while the envs are out of lockstep: like the reference (ns:455-477 samples at reset time) every env takes
    the new links at ITS next reset -- also the envs whose next episode had already been prepared ahead of time in a shadow
    (csrc/pcc_send_restart.hip) from the old parameters: such a shadow is of another generation and must not be swapped in.
    Every env against its own oracle object, all columns and observations, across the change and two more episodes."""
    n, seed, max_steps, T, t_change = 96, 17, 20, 90, 33
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=seed, record_steps=True, auto_reset=True, max_steps=max_steps)
    oenvs = []
    for i in range(n):
        o = oracle.OracleEnv()
        o.rng_philox(seed, i)
        oenvs.append(o)
    obs = env.reset().cpu().numpy()
    assert np.array_equal(obs, np.stack([o.reset() for o in oenvs]).astype(np.float32))
    osteps = np.zeros(n, dtype=int)
    rs = np.random.RandomState(6)
    idx = np.arange(n)
    new = dict(bw=150.0 + idx, dl=0.04 + 0.001 * idx, queue=5.0 + (idx % 40), loss=0.002 * (idx % 10), rate0=90.0 + 2.0 * idx)
    for t in range(T):
        if t < 20 and t % 5 == 2:      # stagger the episode phases (masked resets), so that shadows come into use
            mask = (idx % 4) == ((t // 5) % 4)
            got = env.reset(torch.as_tensor(mask)).cpu().numpy()
            for i in idx[mask]:
                want = oenvs[i].reset()
                osteps[i] = 0
                assert np.array_equal(got[i], want.astype(np.float32)), (t, i)
        if t == t_change:
            env.set_link_params(new["bw"], new["dl"], new["queue"], new["loss"], new["rate0"])
            for i in range(n):
                oenvs[i].set_params(new["bw"][i], new["dl"][i], new["queue"][i], new["loss"][i], [new["rate0"][i]])
        a = rs.uniform(-1, 1.2, n)
        o_gpu, r_gpu, d_gpu, info = env.step(torch.as_tensor(a, device=DEV))
        rows = info["steps"].cpu().numpy()
        o_gpu, d_gpu = o_gpu.cpu().numpy(), d_gpu.cpu().numpy()
        for i in range(n):
            o_ref, r_ref, _, _ = oenvs[i].step(a[i])
            osteps[i] += 1
            done = osteps[i] >= max_steps
            assert np.array_equal(rows[i], oenvs[i].last_row[0]), (t, i)
            assert bool(d_gpu[i]) == done, (t, i)
            if done:
                o_ref = oenvs[i].reset()
                osteps[i] = 0
            assert np.array_equal(o_gpu[i], o_ref.astype(np.float32)), (t, i)
    # every env has restarted since the change: its link is the new one
    assert np.array_equal(env.state("bw").cpu().numpy(), new["bw"])
    stats = env.restart_stats()
    assert stats["shadow_swaps"] > 0, stats          # (the shadows were in use: the test saw what it is about)
    env.check_flags()
    env.close()


def test_two_senders_out_of_lockstep_match_oracle():
    """The same schedule with two senders on the link (restart items run the two-sender wave path and both senders'
    warm-up retires): masked resets, then auto-resets at each env's own episode end, every env against its oracle."""
    n, seed, max_steps, T = 40, 21, 12, 40
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=seed, n_senders=2, record_steps=True, auto_reset=True,
                                       max_steps=max_steps)
    oenvs = []
    for i in range(n):
        o = oracle.OracleEnv(2)
        o.rng_philox(seed, i)
        oenvs.append(o)
    obs = env.reset().cpu().numpy()
    oobs = np.stack([o.reset() for o in oenvs])
    assert np.array_equal(obs, oobs.astype(np.float32))
    osteps = np.zeros(n, dtype=int)
    rs = np.random.RandomState(6)
    idx = np.arange(n)
    for t in range(T):
        if t % 5 == 2 and t < 20:
            mask = (idx % 4) == ((t // 5) % 4)
            got = env.reset(torch.as_tensor(mask)).cpu().numpy()
            for i in idx[mask]:
                want = oenvs[i].reset()
                osteps[i] = 0
                assert np.array_equal(got[i], want.astype(np.float32)), (t, i)
        a = rs.uniform(-1, 1.5, (n, 2))
        o_gpu, r_gpu, d_gpu, info = env.step(torch.as_tensor(a, device=DEV))
        rows = info["steps"].cpu().numpy()
        o_gpu, d_gpu = o_gpu.cpu().numpy(), d_gpu.cpu().numpy()
        for i in range(n):
            o_ref, r_ref, _, _ = oenvs[i].step(a[i])
            osteps[i] += 1
            done = osteps[i] >= max_steps
            assert np.array_equal(rows[i], oenvs[i].last_row), (t, i)
            assert bool(d_gpu[i]) == done, (t, i)
            if done:
                o_ref = oenvs[i].reset()
                osteps[i] = 0
            assert np.array_equal(o_gpu[i], o_ref.astype(np.float32)), (t, i)
    env.check_flags()
    env.close()


@pytest.mark.parametrize("knobs", [dict(), dict(parts=8), dict(parts=8, send_waves=1), dict(parts=8, retire_wide_predict=0.0),
                                   dict(parts=8, retire_sorted=0), dict(light_snake=0, wave_oldest_first=0),
                                   dict(parts=8, prio_level=2, prio_light_items=8, prio_wave_items=8, prio_team=1, team_predict=600.0),
                                   dict(send_waves=1), dict(parts=8, send_waves=32, heavy_item_packets=0.0),
                                   dict(light_front=0), dict(parts=8, light_front=1), dict(parts=8, light_front=3, send_waves=2),
                                   dict(parts=8, light_front=10000, heavy_predict=64.0)])   # (more in front than there are: all of them)
def test_launch_shape_of_the_send_half_does_not_matter(knobs):
    """The send half is one launch with two kinds of workgroup, every workgroup working for one partition of the batch (1 or
    8 of them); which workgroup or wavefront sends or retires an env, in which order and at which priority, must never change
    a result (and every env must be handled exactly once: a dealing that is not a bijection would send an env twice or not
    at all).  3 000 envs with work lists forced on: partitions of 384 envs, the last one partly filled."""
    n_envs, n_steps, seed = 3000, 40, 17
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, record_steps=True, auto_reset=False)
    env.set_tuning(list_min_envs=0, **knobs)
    env.reset()
    acts = np.random.RandomState(seed).uniform(-1, 1.5, (n_envs, n_steps))
    steps, obs, done = run_gpu(env, acts, n_steps)
    ref = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=seed)
    assert np.array_equal(steps, ref["steps"])
    env.close()


def test_two_senders_at_full_size():
    """BASELINE.json configs[4] at its full size (32 768 envs x 2 senders): conservation per sender, the
    queue bound, and the first 256 envs of the batch against the oracle, bit for bit."""
    N, T, M = 32768, 30, 256
    env = pcc_rl_amd.BatchedNetworkEnv(N, device=DEV, seed=4, n_senders=2, record_steps=True, auto_reset=False)
    env.reset()
    gen = torch.Generator(device=DEV).manual_seed(0)
    tail0 = (env.state("acc_tail") + env.state("drop_tail")).clone().long()
    head0 = (env.state("acc_head") + env.state("drop_head")).clone().long()
    sent = torch.zeros((2, N), dtype=torch.float64, device=DEV)
    gone = torch.zeros_like(sent)
    rows_m, acts_m = [], []
    for t in range(T):
        a = torch.rand((N, 2), generator=gen, device=DEV, dtype=torch.float64) * 2.5 - 1
        o, r, d, info = env.step(a)
        s = info["steps"]                      # [N, 2, 19]
        sent += s[:, :, 0].t()
        gone += (s[:, :, 1] + s[:, :, 2]).t()
        rows_m.append(s[:M].clone())
        acts_m.append(a[:M].clone())
        assert bool((env.state("queue_delay") <= env.state("maxq")).all())
    env.check_flags()
    tail = (env.state("acc_tail") + env.state("drop_tail")).long()
    head = (env.state("acc_head") + env.state("drop_head")).long()
    assert bool(((tail - tail0).double() == sent).all())
    assert bool(((head - head0).double() == gone).all())
    ref = oracle.run_batch(torch.stack(acts_m, 1).cpu().numpy(), n_senders=2, rng_mode=oracle.RNG_PHILOX, seed=4, want_obs=False)
    assert np.array_equal(torch.stack(rows_m, 2).cpu().numpy(), ref["steps"])
    env.close()


def test_ring_tiers_promote_and_come_back_at_reset():
    """Rings start in the small tier, envs with many packets in flight are moved up (without
    changing a result: the goldens above cover that, fixed_deepq needs the top tier), reset gives the
    pool rings back, and nothing is ever flagged."""
    n = 2048
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=3, auto_reset=False, ring_pools=(1, 1, 1))   # every env overloads below: worst-case pools
    env.reset()
    assert int(env.state("ring_tier").max().item()) <= 1      # two warm-up MIs rarely need more than the small rings
    gen = torch.Generator(device=DEV).manual_seed(0)
    for t in range(150):
        env.step(torch.rand(n, generator=gen, device=DEV) * 2 - 0.5)   # upward drift: queues fill, drops pile up
    env.check_flags()
    tiers = env.state("ring_tier")[0].cpu().numpy()
    in_flight = (env.state("drop_tail") - env.state("drop_head"))[0].cpu().numpy()
    assert tiers.max() >= 2 and (tiers == 0).any()
    assert (in_flight <= 2 * 512 * 4.0 ** tiers).all()
    env.reset()
    assert int(env.state("ring_tier").max().item()) <= 1
    for t in range(20):
        env.step(torch.rand(n, generator=gen, device=DEV) * 2 - 1)
    env.check_flags()
    env.close()


def test_ring_pool_exhaustion_is_flagged():
    """Too few pool rings for the load: flagged (and the overflow that follows), never silent."""
    n = 4096
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=3, auto_reset=False, ring_pools=(1000000, 1000000, 1000000))   # the minimum: 256 rings per pool
    env.reset()
    gen = torch.Generator(device=DEV).manual_seed(0)
    for t in range(200):
        env.step(torch.rand(n, generator=gen, device=DEV) * 2 - 0.5)
    flags = env.state("flags").cpu().numpy()
    assert (flags & pcc_rl_amd.native.PCC_FLAG_POOL_EXHAUSTED).any()
    with pytest.raises(pcc_rl_amd.PccError):
        env.check_flags()
    env.close()


@pytest.mark.parametrize("name", ["cwnd_pm1", "cwnd_grow", "cwnd_fixed_deepq"])
def test_use_cwnd_goldens_bit_exact(name):
    """The reference's dormant USE_CWND engine option (window-limited sending, 2-D actions) on
    traces of the unmodified reference, incl. its quirk that a blocked SEND still passes through the
    link's queue and loss draw."""
    d = load(name)
    n = d["seed"].shape[0]
    feats = [str(f) for f in d["features"]]
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, history_len=int(d["history_len"]), features=feats,
                                       record_steps=True, auto_reset=False, use_cwnd=True)
    p = d["params"]
    env.set_link_params(p[:, 0], p[:, 1], np.round(p[:, 2]), p[:, 3], p[:, 4])
    k = int((d["rng"][:, 1] - d["rng"][:, 0]).max())
    trace = np.stack([oracle.mt_uniforms(int(s), k, skip=int(o)) for s, o in zip(d["seed"], d["rng"][:, 0])])
    env.set_loss_trace(trace)
    obs0 = env.reset().cpu().numpy()
    assert np.array_equal(obs0, d["obs0"].astype(np.float32))
    assert np.array_equal(env.state("now").cpu().numpy(), d["warm"][:, 0])
    T = d["actions"].shape[1]
    rows, cw = [], []
    for t in range(T):
        o, r, dn, info = env.step(d["actions"][:, t])          # [n, 2]
        rows.append(info["steps"].clone())
        cw.append(env.state("cwnd")[0].clone())
    env.check_flags()
    steps = torch.stack(rows, 1).cpu().numpy()
    assert np.array_equal(torch.stack(cw, 1).cpu().numpy(), d["cwnd"])
    assert np.array_equal(steps[..., :3], d["steps"][..., :3])
    assert np.array_equal(steps, d["steps"])
    env.close()


def test_use_cwnd_philox_batch_matches_oracle():
    n_envs, n_steps, seed = 300, 120, 77
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, record_steps=True, auto_reset=False,
                                       use_cwnd=True)
    env.reset()
    rs = np.random.RandomState(seed)
    acts = rs.uniform(-1, 1, (n_envs, n_steps, 2))
    acts[..., 1] = rs.uniform(-1, 3, (n_envs, n_steps))      # windows drift up: blocked and unblocked phases
    rows, obs = [], []
    for t in range(n_steps):
        o, r, dn, info = env.step(acts[:, t])
        rows.append(info["steps"].clone()); obs.append(o.clone())
    env.check_flags()
    steps = torch.stack(rows, 1).cpu().numpy()
    ref = oracle.run_batch(acts[..., 0], rng_mode=oracle.RNG_PHILOX, seed=seed, cwnd_actions=acts[..., 1])
    assert np.array_equal(steps[..., :3], ref["steps"][..., :3])
    assert np.array_equal(steps, ref["steps"])
    assert np.array_equal(torch.stack(obs, 1).cpu().numpy(), ref["obs"].astype(np.float32))
    env.close()


@pytest.mark.parametrize("name", ["noise_pm1", "noise_fixed_q1", "noise_fixed_lossy", "noise_fixed_deepq"])
def test_use_latency_noise_goldens_bit_exact(name):
    """The reference's dormant USE_LATENCY_NOISE engine option (every link latency x random.uniform(1.0, 1.1):
    packets overtake each other on both hops) on traces of the unmodified reference run with the flag set."""
    d = load(name)
    n = d["seed"].shape[0]
    feats = [str(f) for f in d["features"]]
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, history_len=int(d["history_len"]), features=feats,
                                       record_steps=True, auto_reset=False, latency_noise=1.1)
    p = d["params"]
    env.set_link_params(p[:, 0], p[:, 1], np.round(p[:, 2]), p[:, 3], p[:, 4])
    k = int((d["rng"][:, 1] - d["rng"][:, 0]).max())
    trace = np.stack([oracle.mt_uniforms(int(s), k, skip=int(o)) for s, o in zip(d["seed"], d["rng"][:, 0])])
    env.set_loss_trace(trace)
    obs0 = env.reset().cpu().numpy()
    assert np.array_equal(obs0, d["obs0"].astype(np.float32))
    assert np.array_equal(env.state("now").cpu().numpy(), d["warm"][:, 0])
    rows, obs = [], []
    for t in range(d["actions"].shape[1]):
        o, r, dn, info = env.step(d["actions"][:, t])
        rows.append(info["steps"].clone()); obs.append(o.clone())
    env.check_flags()
    steps = torch.stack(rows, 1).cpu().numpy()
    assert np.array_equal(steps[..., :3], d["steps"][..., :3])
    assert np.array_equal(steps, d["steps"])
    nf = d["obs_tail"].shape[2]
    assert np.array_equal(torch.stack(obs, 1).cpu().numpy()[..., -nf:], d["obs_tail"].astype(np.float32))
    env.close()


@pytest.mark.parametrize("name", ["cwnd_noise_pm1", "cwnd_noise_grow"])
def test_both_engine_options_together_goldens_bit_exact(name):
    """USE_CWND and USE_LATENCY_NOISE switched on together in the unmodified reference (module globals, ns:51-54): the
    event heap of the noise engine with the window test in front of every SEND."""
    d = load(name)
    n = d["seed"].shape[0]
    feats = [str(f) for f in d["features"]]
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, history_len=int(d["history_len"]), features=feats,
                                       record_steps=True, auto_reset=False, latency_noise=1.1, use_cwnd=True)
    p = d["params"]
    env.set_link_params(p[:, 0], p[:, 1], np.round(p[:, 2]), p[:, 3], p[:, 4])
    k = int((d["rng"][:, 1] - d["rng"][:, 0]).max())
    trace = np.stack([oracle.mt_uniforms(int(s), k, skip=int(o)) for s, o in zip(d["seed"], d["rng"][:, 0])])
    env.set_loss_trace(trace)
    obs0 = env.reset().cpu().numpy()
    assert np.array_equal(obs0, d["obs0"].astype(np.float32))
    assert np.array_equal(env.state("now").cpu().numpy(), d["warm"][:, 0])
    rows = []
    for t in range(d["actions"].shape[1]):
        o, r, dn, info = env.step(d["actions"][:, t])
        rows.append(info["steps"].clone())
    env.check_flags()
    steps = torch.stack(rows, 1).cpu().numpy()
    assert np.array_equal(steps[..., :3], d["steps"][..., :3])
    assert np.array_equal(steps, d["steps"])
    assert np.array_equal(env.state("cwnd")[0].cpu().numpy(), d["cwnd"][:, -1])
    env.close()


def test_both_engine_options_together_philox_batch_matches_oracle():
    n_envs, n_steps, seed = 200, 80, 57
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, record_steps=True, auto_reset=False, latency_noise=1.1,
                                       use_cwnd=True)
    env.reset()
    rs = np.random.RandomState(seed)
    acts = np.stack([rs.uniform(-1, 1, (n_envs, n_steps)), rs.uniform(-1, 3, (n_envs, n_steps))], axis=2)
    a = torch.as_tensor(acts, dtype=torch.float64, device=DEV)
    rows = []
    for t in range(n_steps):
        o, r, d, info = env.step(a[:, t])
        rows.append(info["steps"].clone())
    env.check_flags()
    steps = torch.stack(rows, 1).cpu().numpy()
    ref = oracle.run_batch(acts[..., 0], rng_mode=oracle.RNG_PHILOX, seed=seed, cwnd_actions=acts[..., 1], latency_noise=1.1,
                           want_obs=False)
    assert np.array_equal(steps[..., :3], ref["steps"][..., :3])
    assert np.array_equal(steps, ref["steps"])
    env.close()


def test_use_latency_noise_philox_batch_matches_oracle():
    """... and on its own counter-based streams against the oracle with the option on: 200 envs with random links,
    two episodes back to back (auto-reset), every column and the observations."""
    n_envs, n_steps, seed = 200, 60, 91
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, record_steps=True, auto_reset=True,
                                       latency_noise=1.1, max_steps=n_steps // 2)
    env.reset()
    acts = np.random.RandomState(seed).uniform(-1, 1, (n_envs, n_steps))
    rows, obs = [], []
    for t in range(n_steps):
        o, r, dn, info = env.step(acts[:, t])
        rows.append(info["steps"].clone()); obs.append(o.clone())
    env.check_flags()
    steps = torch.stack(rows, 1).cpu().numpy()
    half = n_steps // 2
    obs = torch.stack(obs, 1).cpu().numpy()
    ref1 = oracle.run_batch(acts[:, :half], rng_mode=oracle.RNG_PHILOX, seed=seed, latency_noise=1.1)
    assert np.array_equal(steps[:, :half, :3], ref1["steps"][..., :3])
    assert np.array_equal(steps[:, :half], ref1["steps"])
    assert np.array_equal(obs[:, :half - 1], ref1["obs"].astype(np.float32)[:, :half - 1])
    assert np.array_equal(obs[:, half - 1], ref1["obs0"].astype(np.float32))     # the auto-reset's observation
    # episode index 1 of the same envs (run_batch reports the last of its episodes)
    ref2 = oracle.run_batch(acts[:, half:], rng_mode=oracle.RNG_PHILOX, seed=seed, latency_noise=1.1, n_episodes=2)
    assert np.array_equal(steps[:, half:], ref2["steps"])
    assert np.array_equal(obs[:, half:-1], ref2["obs"].astype(np.float32)[:, :half - 1])
    # the noiseless engine gives other numbers on the same streams (the option really is on)
    plain = oracle.run_batch(acts[:, :half], rng_mode=oracle.RNG_PHILOX, seed=seed)
    assert not np.array_equal(plain["steps"], ref1["steps"])
    env.close()


def test_latency_noise_with_a_latency_below_the_clock_resolution_matches_oracle():
    """A link latency so small that t + latency == t in double (allowed: the ranges only ask for latency > 0): a packet's arrival
    at the return link then ties with its own SEND.  The heap-free noise path leaves such envs to the event loop (it would
    count two draws fewer); every column equal to the oracle's event loop, next to envs with ordinary latencies."""
    n_envs, n_steps, seed = 96, 24, 17
    rs = np.random.RandomState(seed)
    bw = rs.uniform(100, 500, n_envs)
    # (these envs' clocks stay near 0.01-0.05 s -- an interval is half a round trip -- where an ulp is 2e-18 .. 7e-18)
    dl = np.where(np.arange(n_envs) % 3 == 0, 1e-20, np.where(np.arange(n_envs) % 3 == 1, 4e-19, rs.uniform(0.05, 0.5, n_envs)))
    queue = 1.0 + np.floor(np.exp(rs.uniform(0, 4, n_envs)))
    loss = rs.uniform(0, 0.05, n_envs)
    rate0 = rs.uniform(0.3, 1.5, n_envs) * bw
    params = np.stack([bw, dl, queue, loss, rate0], 1)
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, record_steps=True, auto_reset=False, latency_noise=1.1,
                                       link_params=[torch.tensor(c, dtype=torch.float64, device=DEV) for c in (bw, dl, queue, loss, rate0)])
    env.reset()
    acts = rs.uniform(-1, 1, (n_envs, n_steps))
    rows = []
    for t in range(n_steps):
        o, r, dn, info = env.step(acts[:, t])
        rows.append(info["steps"].clone())
    env.check_flags()
    steps = torch.stack(rows, 1).cpu().numpy()
    ref = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=seed, latency_noise=1.1, params=params)
    assert np.array_equal(steps[..., :3], ref["steps"][..., :3])
    assert np.array_equal(steps, ref["steps"])
    env.close()


@pytest.mark.parametrize("option", ["noise", "cwnd"])
def test_engine_options_out_of_lockstep_match_oracle(option):
    """With the dormant engine options the auto-reset of envs that are out of lockstep stays the gated reset
    launches after the step (no wave path / no send half for restart items): masked resets, then every env's
    own episode ends, against per-env oracles."""
    n, seed, max_steps, T = 24, 31, 10, 27
    kw = dict(latency_noise=1.1) if option == "noise" else dict(use_cwnd=True)
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=seed, record_steps=True, auto_reset=True, max_steps=max_steps, **kw)
    oenvs = []
    for i in range(n):
        o = oracle.OracleEnv()
        o.rng_philox(seed, i)
        if option == "noise":
            o.use_latency_noise(True, 1.1)
        else:
            o.use_cwnd(True)
        oenvs.append(o)
    obs = env.reset().cpu().numpy()
    assert np.array_equal(obs, np.stack([o.reset() for o in oenvs]).astype(np.float32))
    osteps = np.zeros(n, dtype=int)
    rs = np.random.RandomState(8)
    idx = np.arange(n)
    for t in range(T):
        if t in (3, 6):
            mask = (idx % 3) == (t // 3 - 1)
            got = env.reset(torch.as_tensor(mask)).cpu().numpy()
            for i in idx[mask]:
                want = oenvs[i].reset()
                osteps[i] = 0
                assert np.array_equal(got[i], want.astype(np.float32)), (t, i)
        a = rs.uniform(-1, 1.5, (n, 2) if option == "cwnd" else n)
        o_gpu, r_gpu, d_gpu, info = env.step(torch.as_tensor(a, device=DEV))
        rows = info["steps"].cpu().numpy()
        o_gpu, d_gpu = o_gpu.cpu().numpy(), d_gpu.cpu().numpy()
        for i in range(n):
            o_ref, r_ref, _, _ = oenvs[i].step(a[i])
            osteps[i] += 1
            done = osteps[i] >= max_steps
            assert np.array_equal(rows[i], oenvs[i].last_row[0]), (t, i)
            assert bool(d_gpu[i]) == done, (t, i)
            if done:
                o_ref = oenvs[i].reset()
                osteps[i] = 0
            assert np.array_equal(o_gpu[i], o_ref.astype(np.float32)), (t, i)
    env.check_flags()
    env.close()


@pytest.mark.parametrize("name,cwnd,noise", [("two_sender_cwnd", True, False), ("two_sender_noise", False, True),
                                             ("two_sender_cwnd_noise", True, True)])
def test_engine_options_with_two_senders_goldens_bit_exact(name, cwnd, noise):
    """The reference's dormant flags with two senders on the bottleneck (the flags are module globals the engine reads for
    whatever senders it holds, ns:51-54): every sender its own window and [rate action, cwnd action]; events of equal time
    in (sender id, ACK before SEND) order.  The event-loop build, replaying the reference's own uniform stream."""
    d = load(name)
    d["features"] = np.array(pcc_rl_amd.DEFAULT_FEATURES.split(","))
    n = d["seed"].shape[0]
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, n_senders=2, record_steps=True, auto_reset=False, use_cwnd=cwnd,
                                       latency_noise=1.1 if noise else None)
    p = d["params"]
    env.set_link_params(p[:, 0], p[:, 1], np.round(p[:, 2]), p[:, 3], p[:, 4:6])
    k = int((d["rng"][:, 1] - d["rng"][:, 0]).max())
    env.set_loss_trace(np.stack([oracle.mt_uniforms(int(sd), k, skip=int(o)) for sd, o in zip(d["seed"], d["rng"][:, 0])]))
    env.reset()
    assert np.array_equal(env.state("now").cpu().numpy(), d["warm"][:, 0])
    acts = np.stack([d["actions"], d["cwnd_actions"]], axis=3) if cwnd else d["actions"]      # [case, step, sender(, 2)]
    a = torch.as_tensor(acts, dtype=torch.float64, device=DEV)
    rows, obs, cw = [], [], []
    for t in range(acts.shape[1]):
        o, r, dn, info = env.step(a[:, t])
        rows.append(info["steps"].clone()); obs.append(o.clone())
        cw.append(env.state("cwnd").clone())
    env.check_flags()
    steps = torch.stack(rows, 2).cpu().numpy()
    assert np.array_equal(steps[..., :3], d["steps"][..., :3])
    assert np.array_equal(steps, d["steps"])
    assert np.array_equal(torch.stack(obs, 2).cpu().numpy()[..., -3:], d["obs_tail"].astype(np.float32))
    if cwnd:
        assert np.array_equal(torch.stack(cw, 0).permute(2, 0, 1).cpu().numpy(), d["cwnd"])      # [case, step, sender]
    env.close()


@pytest.mark.parametrize("cwnd,noise", [(True, False), (False, True), (True, True)])
def test_engine_options_with_two_senders_philox_batches_match_oracle(cwnd, noise):
    """... and on the device's own Philox stream against the oracle, over an episode boundary (auto-reset out of lockstep:
    the gated reset launches of the event-loop build)."""
    n_envs, n_steps, seed = 150, 60, 91
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, n_senders=2, record_steps=True, use_cwnd=cwnd,
                                       latency_noise=1.1 if noise else None, max_steps=n_steps // 2)
    env.reset()
    rs = np.random.RandomState(seed)
    rate_a, cwnd_a = rs.uniform(-1, 1, (n_envs, n_steps, 2)), rs.uniform(-1, 3, (n_envs, n_steps, 2))
    acts = np.stack([rate_a, cwnd_a], axis=3) if cwnd else rate_a
    a = torch.as_tensor(acts, dtype=torch.float64, device=DEV)
    rows = []
    for t in range(n_steps):
        o, r, dn, info = env.step(a[:, t])
        rows.append(info["steps"].clone())
        assert bool(dn.all()) == (t % (n_steps // 2) == n_steps // 2 - 1)
    env.check_flags()
    steps = torch.stack(rows, 2).cpu().numpy()
    half = n_steps // 2
    kw = dict(n_senders=2, rng_mode=oracle.RNG_PHILOX, seed=seed, latency_noise=1.1 if noise else None, want_obs=False)
    ref1 = oracle.run_batch(rate_a[:, :half], cwnd_actions=cwnd_a[:, :half] if cwnd else None, **kw)
    assert np.array_equal(steps[:, :, :half, :3], ref1["steps"][..., :3])
    assert np.array_equal(steps[:, :, :half], ref1["steps"])
    ref2 = oracle.run_batch(rate_a[:, half:], cwnd_actions=cwnd_a[:, half:] if cwnd else None, n_episodes=2, **kw)
    assert np.array_equal(steps[:, :, half:], ref2["steps"])
    plain = oracle.run_batch(rate_a[:, :half], n_senders=2, rng_mode=oracle.RNG_PHILOX, seed=seed, want_obs=False)
    assert not np.array_equal(plain["steps"], ref1["steps"])       # the options really are on
    env.close()


@pytest.mark.parametrize("scale,n_envs,n_steps", [(1.0, 384, 70), (8.0, 256, 90)])
def test_two_senders_latency_noise_without_the_event_loop_matches_oracle(scale, n_envs, n_steps):
    """Two senders with USE_LATENCY_NOISE run their intervals by sorting too (round 6: noise_sorted2_kernel -- the two SEND streams
    merged by (time, sender) on one draw stream, the sender id ahead of the kind of event in every comparison, blocks of SENDs cut by
    time; the numpy study is tests/models/noise_sorting2_model.py).  Random links, both senders' counts, clocks, rewards, every metric and
    the observations against the oracle's event loop; large actions push the rates from the floor to the ceiling, so that intervals
    of many hundred SENDs -- run as sub-intervals -- are in the comparison.  (tests/test_variants.py runs this with the event loop for
    every env and with the two crossed from env to env.)"""
    seed = 23
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, n_senders=2, record_steps=True, auto_reset=False, latency_noise=1.1)
    obs0 = env.reset().clone()
    acts = np.random.RandomState(seed).uniform(-1, 1, (n_envs, n_steps, 2)) * scale
    a = torch.as_tensor(acts, dtype=torch.float64, device=DEV)
    rows, obs = [], []
    for t in range(n_steps):
        o, r, dn, info = env.step(a[:, t])
        rows.append(info["steps"].clone()); obs.append(o.clone())
    env.check_flags()
    steps = torch.stack(rows, 2).cpu().numpy()                     # [N, S, T, 19]
    got_obs = torch.stack(obs, 2).cpu().numpy()                    # [N, S, T, HF]
    ref = oracle.run_batch(acts, n_senders=2, rng_mode=oracle.RNG_PHILOX, seed=seed, latency_noise=1.1)
    assert np.array_equal(obs0.cpu().numpy().reshape(ref["obs0"].shape), ref["obs0"].astype(np.float32))
    assert np.array_equal(steps[..., :3], ref["steps"][..., :3])
    assert np.array_equal(steps, ref["steps"])
    assert np.array_equal(got_obs.reshape(ref["obs"].shape), ref["obs"].astype(np.float32))
    assert float(steps[..., 0].max()) >= (400 if scale > 1 else 100)   # (intervals of many SENDs are in the batch)
    env.close()


def test_event_loop_build_has_no_send_half():
    env = pcc_rl_amd.BatchedNetworkEnv(8, device=DEV, n_senders=2, use_cwnd=True, auto_reset=False)
    env.reset()
    with pytest.raises(pcc_rl_amd.PccError):
        env.step_send(torch.zeros(8, 2, 2, device=DEV))
    env.close()
    env = pcc_rl_amd.BatchedNetworkEnv(8, device=DEV, latency_noise=1.1, auto_reset=False)
    env.reset()
    with pytest.raises(pcc_rl_amd.PccError):
        env.step_send(torch.zeros(8, device=DEV))
    env.close()


def test_grouped_env_is_the_same_envs_on_several_streams():
    """GroupedNetworkEnv: groups stepped on their own streams give the numbers of one batch."""
    n, T, seed = 512, 80, 9
    one = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=seed, auto_reset=False)
    grp = pcc_rl_amd.GroupedNetworkEnv(n, 4, device=DEV, seed=seed, auto_reset=False)
    o1 = one.reset().clone()
    og = grp.reset()
    assert torch.equal(o1, og)
    gen = torch.Generator(device=DEV).manual_seed(1)
    acts = torch.rand((T, n), generator=gen, device=DEV) * 2 - 1
    ref_obs, ref_rew = [], []
    for t in range(T):
        o, r, d, _ = one.step(acts[t])
        ref_obs.append(o.clone()); ref_rew.append(r.clone())
    got_obs = [[None] * 4 for _ in range(T)]
    got_rew = [[None] * 4 for _ in range(T)]
    m = grp.group_size
    for g in range(4):                      # each group runs ahead on its own: no lock step between groups
        for t in range(T):
            o, r, d, _ = grp.step_group(g, acts[t, g * m:(g + 1) * m])
            with torch.cuda.stream(grp.streams[g]):
                got_obs[t][g] = o.clone(); got_rew[t][g] = r.clone()
    grp.synchronize()
    grp.check_flags()
    for t in range(T):
        assert torch.equal(torch.cat(got_obs[t], 0), ref_obs[t]), t
        assert torch.equal(torch.cat(got_rew[t], 0), ref_rew[t]), t
    one.close(); grp.close()


def test_queue_limits_just_above_a_power_of_two_match_oracle():
    """Regime C of the wave passes: envs whose queue limit sits just above a power of two (0.25 .. 16 s) and whose sender
    overdrives the link -- the full queue straddles the power of two, fl(qcur + 1/bw) rounds on two grids -- against the
    oracle, every column; half of them sent one env per wavefront from the first interval on."""
    n_envs, n_steps, seed = 512, 160, 41
    rs = np.random.RandomState(seed)
    bw = rs.uniform(100, 500, n_envs)
    B = 2.0 ** rs.randint(-2, 5, n_envs)
    queue = np.maximum(2.0, np.ceil(B * bw + rs.uniform(0, 0.9, n_envs)))
    dl = rs.uniform(0.05, 0.5, n_envs)
    loss = np.where(rs.rand(n_envs) < 0.2, 0.0, rs.uniform(0, 0.05, n_envs))
    rate0 = np.minimum(1000.0, bw * rs.uniform(1.05, 1.6, n_envs))
    acts = rs.uniform(-0.4, 1.0, (n_envs, n_steps))
    ref = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=seed, params=np.stack([bw, dl, queue, loss, rate0], 1))
    for knobs in (dict(), dict(heavy_predict=0.0, team_predict=1e18, heavy_item_packets=0.0)):
        env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, record_steps=True, auto_reset=False,
                                           link_params=(bw, dl, queue, loss, rate0))
        env.set_tuning(**knobs)
        env.reset()
        steps, obs, done = run_gpu(env, acts, n_steps)
        assert np.array_equal(steps[..., :3], ref["steps"][..., :3]), knobs
        assert np.array_equal(steps, ref["steps"]), knobs
        assert np.array_equal(obs, ref["obs"].astype(np.float32)), knobs
        env.close()


@pytest.mark.parametrize("case", ["overdriven", "edges", "mixed"])
def test_two_sender_token_pass_matches_oracle(case):
    """The closed-form pass of the two-sender wave path (heavy_mi2: backlogged queue in one binade, accept decisions by a
    token scan over the merged stream with uneven token arrivals) against the oracle, every env sent by the wave path
    from its first packet: two senders that together overdrive the link (the pass's home ground), queue limits around
    powers of two and 1/bw on both sides of the queue's binade (its preconditions break and hold by turns), and the
    default parameter ranges."""
    n_envs, n_steps, seed = 768, 120, {"overdriven": 61, "edges": 62, "mixed": 63}[case]
    rs = np.random.RandomState(seed)
    bw = rs.uniform(100, 500, n_envs)
    dl = rs.uniform(0.05, 0.5, n_envs)
    loss = np.where(rs.rand(n_envs) < 0.25, 0.0, rs.uniform(0, 0.05, n_envs))
    if case == "edges":
        B = 2.0 ** rs.randint(-3, 5, n_envs)
        queue = np.maximum(2.0, np.ceil(B * bw + rs.uniform(-3.0, 3.0, n_envs)))
    else:
        queue = 1.0 + np.floor(np.exp(rs.uniform(0, 8, n_envs)))
    share = rs.uniform(0.2, 0.8, n_envs)
    load = rs.uniform(1.02, 1.8, n_envs) if case != "mixed" else rs.uniform(0.4, 1.6, n_envs)
    rate0 = np.clip(np.stack([bw * load * share, bw * load * (1.0 - share)], 1), 40.0, 1000.0)
    acts = rs.uniform(-0.5, 0.8, (n_envs, n_steps, 2))
    ref = oracle.run_batch(acts, n_senders=2, rng_mode=oracle.RNG_PHILOX, seed=seed,
                           params=np.concatenate([np.stack([bw, dl, queue, loss], 1), rate0], 1))
    for knobs in (dict(heavy_predict=0.0), dict(takeover_lanes=64, round_packets=8)):
        env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, n_senders=2, record_steps=True, auto_reset=False,
                                           link_params=(bw, dl, queue, loss, rate0))
        env.set_tuning(**knobs)
        env.reset()
        steps, obs, done = run_gpu(env, acts, n_steps)
        bad = np.argwhere((steps[..., :3] != ref["steps"][..., :3]).any(axis=(1, 2, 3)))
        assert bad.size == 0, (knobs, "envs with count mismatches: %s" % bad[:10].ravel())
        assert np.array_equal(steps, ref["steps"]), knobs
        assert np.array_equal(obs, ref["obs"].astype(np.float32)), knobs
        env.close()


@pytest.mark.parametrize("n_senders", [1, 2])
def test_small_batch_path_without_work_lists(n_senders):
    """Batches below 8192 envs are stepped in index order, without work lists (the library's default, which the other
    tests switch off): two episodes with the auto-reset between them against the oracle, every column; then a masked
    reset takes the batch out of lockstep and the following auto-resets (gated reset launches here, not restart items)
    must leave every env in its own episode."""
    pcc_rl_amd.BatchedNetworkEnv.DEFAULT_LIST_MIN_ENVS = None
    n_envs, seed, T = 300, 9, 20
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, n_senders=n_senders, record_steps=True, auto_reset=True,
                                       max_steps=T)
    env.reset()
    rs = np.random.RandomState(seed)
    shape = (n_envs, 2 * T) if n_senders == 1 else (n_envs, 2 * T, 2)
    acts = rs.uniform(-1, 1.5, shape)
    steps, obs, done = run_gpu(env, acts, 2 * T)
    ref1 = oracle.run_batch(acts[:, :T], n_senders=n_senders, rng_mode=oracle.RNG_PHILOX, seed=seed, want_obs=False)
    ref2 = oracle.run_batch(acts[:, T:], n_senders=n_senders, rng_mode=oracle.RNG_PHILOX, seed=seed, n_episodes=2, want_obs=False)
    assert np.array_equal(steps[..., :T, :], ref1["steps"])
    assert np.array_equal(steps[..., T:, :], ref2["steps"])
    # out of lockstep: every third env restarts now, everybody steps on through two more boundaries
    mask = torch.arange(n_envs, device=DEV) % 3 == 0
    env.reset(mask)
    for t in range(2 * T + 3):
        env.step(torch.zeros((n_envs, n_senders), device=DEV))
    torch.cuda.synchronize()
    env.check_flags()
    st = env.state("steps").cpu().numpy()
    ep = env.state("episode").cpu().numpy()
    m = mask.cpu().numpy()
    assert (st[m] == 3).all() and (st[~m] == 3).all()          # (2T + 3) steps after a boundary everybody shares modulo T
    assert (ep[~m] == ep[~m][0]).all() and (ep[m] == ep[~m][0] + 1).all()      # the masked envs are one episode ahead
    env.close()


def test_long_episode_matches_oracle():
    """The near-group tolerance (1e-12 relative) is an assumption about the size of the clock: 20 000-step episodes
    (clocks up to ~1e5 s, 50 x the default episode) still match the oracle bit for bit, and no env raises
    PCC_FLAG_TIME_RANGE inside the default parameter ranges."""
    n_envs, n_steps, seed = 6, 20000, 77
    env = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, record_steps=True, auto_reset=False, max_steps=n_steps)
    env.reset()
    rs = np.random.RandomState(seed)
    acts = rs.uniform(-1, 1, (n_envs, n_steps))
    a = torch.as_tensor(acts, dtype=torch.float64, device=DEV)
    rows = torch.empty((n_steps, n_envs, 19), dtype=torch.float64, device=DEV)
    for t in range(n_steps):
        o, r, d, info = env.step(a[:, t])
        rows[t] = info["steps"]
    torch.cuda.synchronize()
    env.check_flags()
    ref = oracle.run_batch(acts, rng_mode=oracle.RNG_PHILOX, seed=seed, want_obs=False)
    got = rows.permute(1, 0, 2).cpu().numpy()
    assert np.array_equal(got[..., :3], ref["steps"][..., :3])
    assert np.array_equal(got, ref["steps"])
    assert float(got[..., 4].max()) > 2e4      # the clocks did get large
    env.close()


def test_bad_inputs_are_flagged_and_nothing_hangs():
    """A NaN action is applied as 0 and flagged (PCC_FLAG_BAD_ACTION) -- the env goes on being stepped and filed; a
    starting rate of 0 or NaN is flagged (PCC_FLAG_BAD_PARAMS) and the env runs on a stand-in link instead of spinning;
    a clock beyond what the dropped packets' ordering tolerates raises PCC_FLAG_TIME_RANGE."""
    from pcc_rl_amd import native
    n = 256
    env = pcc_rl_amd.BatchedNetworkEnv(n, device=DEV, seed=5, auto_reset=False, record_steps=True)
    env.reset()
    a = torch.zeros(n, dtype=torch.float64, device=DEV)
    a[7] = float("nan")
    rate_before = env.state("rate")[0].clone()
    for t in range(5):
        o, r, d, info = env.step(a)
    torch.cuda.synchronize()
    flags = env.state("flags").cpu().numpy()
    assert flags[7] == native.PCC_FLAG_BAD_ACTION and (np.delete(flags, 7) == 0).all()
    assert torch.equal(env.state("rate")[0], rate_before.clamp(40.0, 1000.0))     # zero (and NaN -> zero) actions: only the clamp
    assert int(env.state("steps")[7].item()) == 5 and bool(torch.isfinite(o).all())
    env.close()
    env = pcc_rl_amd.BatchedNetworkEnv(4, device=DEV, seed=5, auto_reset=False)
    env.set_link_params(200.0, 0.03, 5.0, 0.0, torch.tensor([60.0, 0.0, float("nan"), -3.0], dtype=torch.float64))
    env.reset()
    for t in range(3):
        env.step(torch.zeros(4, device=DEV))
    torch.cuda.synchronize()
    flags = env.state("flags").cpu().numpy()
    assert flags[0] == 0 and (flags[1:] == native.PCC_FLAG_BAD_PARAMS).all()
    env.close()
    env = pcc_rl_amd.BatchedNetworkEnv(2, device=DEV, seed=5, auto_reset=False, max_steps=100000)
    env.set_link_params(1e6, 100.0, 5.0, 0.0, 40.0)       # 1/bw = 1e-6 s, an RTT of 200 s: the clock passes 15 625 s quickly
    env.reset()
    for t in range(400):
        env.step(torch.full((2,), -1.0, device=DEV))
    torch.cuda.synchronize()
    flags = env.state("flags").cpu().numpy()
    assert float(env.state("now").max().item()) > 16000.0
    assert (flags & native.PCC_FLAG_TIME_RANGE).all()
    env.close()


def test_parameter_ranges_are_validated():
    """What the exact formulation rests on (a physical link) is checked, not assumed: bad sampling ranges are
    refused by the C ABI, bad caller-supplied link arrays are flagged per env at reset."""
    env = pcc_rl_amd.BatchedNetworkEnv(64, device=DEV, seed=1)
    for lo, hi in [((0.0, 0.05, 0, 0.0, 0.3), (500, 0.5, 8, 0.05, 1.5)),      # bandwidth may be 0
                   ((100, 0.0, 0, 0.0, 0.3), (500, 0.5, 8, 0.05, 1.5)),       # latency may be 0
                   ((100, 0.05, 0, 0.0, 0.3), (500, 0.5, 8, 1.5, 1.5)),       # loss probability > 1
                   ((100, 0.05, 0, 0.0, 0.3), (2e8, 0.5, 8, 0.05, 1.5))]:     # 1/bw below 1e-8 s
        with pytest.raises(pcc_rl_amd.PccError):
            env.randomize_link_params((lo, hi))
    env.randomize_link_params(((100, 0.05, 0, 0.0, 0.3), (500, 0.5, 8, 0.05, 1.5)))
    env.reset()
    env.check_flags()
    bw = torch.full((64,), 200.0, dtype=torch.float64)
    bw[5] = 0.0
    env.set_link_params(bw, 0.03, 5.0, 0.0, 60.0)
    env.reset()
    flags = env.state("flags").cpu().numpy()
    assert flags[5] & 16 and not (np.delete(flags, 5) & 16).any()
    with pytest.raises(pcc_rl_amd.PccError):
        env.check_flags()
    env.close()


@pytest.mark.parametrize("n_envs,lists", [(512, False), (512, True)])
def test_step_many_equals_single_steps(n_envs, lists):
    """pcc_step_many (the loop over the steps in C) gives the rows of the same number of pcc_step calls, episode boundary
    and auto-reset included, on the small-batch path (one launch per step) and with work lists."""
    if not lists:
        pcc_rl_amd.BatchedNetworkEnv.DEFAULT_LIST_MIN_ENVS = None
    T, seed = 50, 17
    acts = torch.as_tensor(np.random.RandomState(seed).uniform(-1, 1.5, (T, n_envs)), dtype=torch.float32, device=DEV)
    one = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, max_steps=20, record_steps=True)
    many = pcc_rl_amd.BatchedNetworkEnv(n_envs, device=DEV, seed=seed, max_steps=20)
    assert torch.equal(one.reset(), many.reset())
    obs = torch.empty((T, n_envs, one.obs_dim), device=DEV)
    rew = torch.empty((T, n_envs), device=DEV)
    done = torch.empty((T, n_envs), dtype=torch.uint8, device=DEV)
    rows = torch.empty((T, n_envs, 1, 19), dtype=torch.float64, device=DEV)
    many.step_many(acts, obs, rew, done, rows)
    for t in range(T):
        o, r, d, info = one.step(acts[t])
        assert torch.equal(o, obs[t]) and torch.equal(r, rew[t]) and torch.equal(d, done[t].view(torch.bool)), t
        assert torch.equal(info["steps"], rows[t, :, 0]), t
    assert int(done.sum().item()) == 2 * n_envs
    one.check_flags(); many.check_flags()
    assert torch.equal(one.state("now"), many.state("now"))
    one.close(); many.close()
