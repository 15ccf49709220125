"""The PPO caller (pcc-rl_amd/ppo.py): advantage recursion on CPU, a short training run on the GPU."""
import numpy as np
import pytest
import torch

from pcc_rl_amd.ppo import MlpPolicy, gae


def test_gae_matches_plain_loop():
    rs = np.random.RandomState(0)
    T, N = 13, 5
    r, v = rs.randn(T, N), rs.randn(T, N)
    d = rs.rand(T, N) < 0.2
    last = rs.randn(N)
    adv, ret = gae(torch.tensor(r), torch.tensor(v), torch.tensor(d), torch.tensor(last), 0.99, 0.95)
    want = np.zeros((T, N))
    for n in range(N):
        run, nxt = 0.0, last[n]
        for t in range(T - 1, -1, -1):
            alive = 0.0 if d[t, n] else 1.0
            delta = r[t, n] + 0.99 * nxt * alive - v[t, n]
            run = delta + 0.99 * 0.95 * alive * run
            want[t, n] = run
            nxt = v[t, n]
    assert np.allclose(adv.numpy(), want, rtol=1e-12, atol=1e-12)
    assert np.allclose(ret.numpy(), want + v, rtol=1e-12, atol=1e-12)


def test_policy_shapes():
    pol = MlpPolicy(30, 1, (32, 16))
    a, logp, v = pol.act(torch.zeros(7, 30))
    assert a.shape == (7, 1) and logp.shape == (7,) and v.shape == (7,)
    # layer sizes of the reference's --arch default (src/gym/stable_solve.py:30)
    assert [m.out_features for m in pol.pi if isinstance(m, torch.nn.Linear)] == [32, 16, 1]


@pytest.mark.gpu
def test_short_training_run_improves_reward():
    """Whole-episode iterations (horizon = 400 steps) so every iteration averages the same mix of
    episode phases; a handful of PPO iterations must raise the mean reward."""
    import pcc_rl_amd
    from pcc_rl_amd.ppo import PPO
    env = pcc_rl_amd.BatchedNetworkEnv(1024, device="cuda:0", seed=3)
    agent = PPO(env, horizon=400, seed=0)
    rewards = [agent.iterate()["mean_step_reward"] for _ in range(6)]
    env.check_flags()
    assert all(np.isfinite(r) for r in rewards)
    assert max(rewards[3:]) > rewards[0], rewards
