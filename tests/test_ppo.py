"""The PPO caller (pcc-rl_amd/ppo.py): advantage recursion on CPU, a short training run on the GPU."""
import numpy as np
import pytest
import torch

from pcc_rl_amd.ppo import MlpPolicy, gae, ppo_loss


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


def test_ppo_objective_matches_plain_loop():
    """The clipped surrogate, value and entropy terms of one minibatch against a per-sample numpy loop
    (fixed seed, float64)."""
    torch.manual_seed(3)
    pol = MlpPolicy(6, 1, (8, 4)).double()
    with torch.no_grad():
        pol.log_std.fill_(-0.3)
    rs = np.random.RandomState(1)
    B = 17
    obs, act = torch.tensor(rs.randn(B, 6)), torch.tensor(rs.randn(B, 1))
    logp_old, adv, ret = torch.tensor(rs.randn(B) * 0.3 - 1.0), torch.tensor(rs.randn(B)), torch.tensor(rs.randn(B))
    loss, pg, vf, ent = ppo_loss(pol, obs, act, logp_old, adv, ret, clip=0.2, ent_coef=0.01)
    mu = pol.pi(obs).detach().numpy()[:, 0]
    val = pol.vf(obs).detach().numpy()[:, 0]
    sd = float(np.exp(-0.3))
    want_pg = want_vf = 0.0
    for b in range(B):
        a = act[b, 0].item()
        logp = -0.5 * ((a - mu[b]) / sd) ** 2 - np.log(sd) - 0.5 * np.log(2 * np.pi)
        ratio = np.exp(logp - logp_old[b].item())
        A = adv[b].item()
        want_pg += -min(ratio * A, min(max(ratio, 0.8), 1.2) * A) / B
        want_vf += 0.5 * (val[b] - ret[b].item()) ** 2 / B
    want_ent = 0.5 + 0.5 * np.log(2 * np.pi) + np.log(sd)
    assert np.isclose(pg.item(), want_pg, rtol=1e-12) and np.isclose(vf.item(), want_vf, rtol=1e-12)
    assert np.isclose(ent.item(), want_ent, rtol=1e-12)
    assert np.isclose(loss.item(), want_pg + want_vf - 0.01 * want_ent, rtol=1e-12)


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


@pytest.mark.gpu
def test_fused_policy_forward_matches_the_framework_path():
    """pcc_policy_act (one launch: mean, sample, log-probability, value) against torch's fp32 evaluation
    of the same networks; tolerance 1e-5 absolute on O(1) values (fp32 sums in a different order)."""
    torch.manual_seed(1)
    dev = torch.device("cuda:0")
    pol = MlpPolicy(30, 1, (32, 16)).to(dev)
    with torch.no_grad():
        pol.log_std.fill_(-0.7)
    obs = torch.randn(5000, 30, device=dev)
    a, logp, v = pol.act_fused(obs, stochastic=False)
    mu = pol.pi(obs).detach()
    assert a.shape == (5000, 1) and torch.allclose(a, mu, atol=1e-5)
    assert torch.allclose(v, pol.value(obs).detach(), atol=1e-5)
    d = pol.dist(obs)
    assert torch.allclose(logp, d.log_prob(mu).sum(-1).detach(), atol=1e-5)
    a2, logp2, v2 = pol.act_fused(obs, stochastic=True)
    assert torch.allclose(logp2, d.log_prob(a2).sum(-1).detach(), atol=1e-4)
    z = (a2 - mu) / pol.log_std.exp()
    assert abs(float(z.mean())) < 0.06 and abs(float(z.std()) - 1.0) < 0.06      # standard-normal draws
    pol7 = MlpPolicy(7, 1, (32, 16)).to(dev)
    a3, _, _ = pol7.act_fused(torch.randn(64, 7, device=dev))                       # no instantiation: framework path
    assert a3.shape == (64, 1)
