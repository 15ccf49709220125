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
def test_short_training_run_beats_the_random_policy_by_a_margin():
    """Whole-episode iterations (horizon = 400 steps) so every iteration averages the same mix of episode phases.  The first
    iterations ARE the random policy (N(0, 1) actions around a zero-initialised mean: ~300 per episode); ten iterations at
    1 024 envs (4.1e6 env-steps, the reference's whole budget) must lift the episode return by half of that at least
    (measured: ~295 -> ~615; profiles/r05_learning_curve.json has the long run: 289 -> 814 in 2e8 env-steps, and the trained
    controller against baselines on held-out envs)."""
    import pcc_rl_amd
    from pcc_rl_amd.ppo import PPO
    env = pcc_rl_amd.BatchedNetworkEnv(1024, device="cuda:0", seed=3)
    agent = PPO(env, horizon=400, seed=0)
    returns = [agent.iterate()["mean_step_reward"] * env.max_steps for _ in range(10)]
    env.check_flags()
    assert all(np.isfinite(r) for r in returns)
    random_policy = sum(returns[:2]) / 2
    assert sum(returns[-3:]) / 3 > 1.5 * random_policy, returns


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


def test_share_flat_makes_the_parameters_views_of_one_tensor():
    torch.manual_seed(2)
    pol = MlpPolicy(30, 1, (32, 16))
    obs = torch.randn(9, 30)
    before = (pol.pi(obs).detach().clone(), pol.value(obs).detach().clone())
    flat = pol.share_flat()
    assert torch.equal(flat, pol.flat_params()) and flat.numel() == 2 * (30 * 32 + 32 + 32 * 16 + 16 + 16 + 1) + 1
    assert torch.equal(pol.pi(obs).detach(), before[0]) and torch.equal(pol.value(obs).detach(), before[1])
    with torch.no_grad():
        flat.mul_(0.5)                                         # an in-place optimiser step on the flat tensor ...
    assert torch.equal(flat, pol.flat_params())                # ... is a step on the module's parameters
    assert not torch.equal(pol.pi(obs).detach(), before[0])
    loss = pol.dist(obs).log_prob(torch.zeros(9, 1)).sum() + pol.value(obs).sum()   # autograd still reaches every parameter
    loss.backward()
    assert all(p.grad is not None for p in pol.parameters())


def _rollout_like(n, D, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    obs = torch.randn(n, D, generator=g).to(dev)
    act = (0.5 * torch.randn(n, 1, generator=g)).to(dev)
    logp = (-1.0 + 0.3 * torch.randn(n, generator=g)).to(dev)
    adv = torch.randn(n, generator=g).to(dev)
    adv[::97] = 0.0
    ret = (2.0 * torch.randn(n, generator=g)).to(dev)
    return obs, act, logp, adv, ret


class _Env(object):   # what PPO.__init__ needs of an env, without a simulator behind it
    def __init__(self, D, dev):
        self.obs_dim, self.device, self.n_senders, self.n_envs = D, dev, 1, 8

    def reset(self):
        return torch.zeros(self.n_envs, self.obs_dim, device=self.device)


@pytest.mark.gpu
@pytest.mark.parametrize("D,n,count", [(30, 5000, 3333), (30, 70000, 70000), (12, 1000, 64), (3, 777, 700)])
def test_fused_minibatch_gradient_matches_autograd(D, n, count):
    """pcc_ppo_minibatch_step's gradient (fp32 FMAs + fp32 MFMA sums over the samples) against torch autograd of ppo_loss in
    float64 on the same minibatch: within 1e-5 of the gradient's largest entry (fp32 sums of `count` terms)."""
    from pcc_rl_amd.ppo import PPO, ppo_loss
    dev = torch.device("cuda:0")
    agent = PPO(_Env(D, dev), seed=5)
    assert agent.fused_update
    with torch.no_grad():
        agent.policy.log_std.fill_(-0.4)
    obs, act, logp, adv, ret = _rollout_like(n, D, dev, 11)
    with torch.no_grad():                                     # log-probabilities near the policy's own: ratios around 1,
        logp = agent.policy.dist(obs).log_prob(act).sum(-1) + 0.15 * torch.randn(n, device=dev)   # some of them clipped
    perm = torch.randperm(n, device=dev)
    g = torch.zeros_like(agent.flat)
    start = min(5, n - count)
    agent.minibatch_step_fused(obs, act.reshape(n), logp, adv, ret, perm, start, count, lr=0.0, grad_out=g)
    stats = agent.stats_buf.tolist()
    idx = perm[start:start + count]
    pol64 = MlpPolicy(D, 1, (32, 16)).to(dev).double()
    pol64.load_state_dict({k: v.double() for k, v in agent.policy.state_dict().items()})
    loss, pg, vf, ent = ppo_loss(pol64, obs[idx].double(), act[idx].double(), logp[idx].double(), adv[idx].double(),
                                 ret[idx].double(), agent.clip, agent.ent_coef)
    loss.backward()
    def net(seq):
        return [p.grad.reshape(-1) for m in seq if isinstance(m, torch.nn.Linear) for p in (m.weight, m.bias)]
    want = torch.cat(net(pol64.pi) + [pol64.log_std.grad.reshape(-1)] + net(pol64.vf))
    err = (g.double() - want).abs().max().item()
    assert err <= 1e-5 * want.abs().max().item() + 1e-7, (err, want.abs().max().item())
    assert abs(-stats[0] - float(pg)) < 1e-4 * max(1.0, abs(float(pg))) and abs(0.5 * stats[1] - float(vf)) < 1e-4 * max(1.0, float(vf))
    assert 0.0 < stats[2] < 1.0                               # some ratios were clipped, not all


@pytest.mark.gpu
def test_fused_optimiser_steps_match_torch_adam():
    """Three consecutive fused steps (gradient + Adam on the shared flat parameters) against autograd + torch.optim.Adam from
    the same start on the same minibatches."""
    from pcc_rl_amd.ppo import PPO, ppo_loss
    dev = torch.device("cuda:0")
    agent = PPO(_Env(30, dev), seed=7)
    ref = MlpPolicy(30, 1, (32, 16)).to(dev)
    ref.load_state_dict(agent.policy.state_dict())
    opt = torch.optim.Adam(ref.parameters(), lr=agent.lr, eps=agent.adam_eps)
    n = 4096
    obs, act, logp, adv, ret = _rollout_like(n, 30, dev, 3)
    with torch.no_grad():
        logp = ref.dist(obs).log_prob(act).sum(-1) + 0.1 * torch.randn(n, device=dev)
    for k in range(3):
        agent.minibatch_step_fused(obs, act.reshape(n), logp, adv, ret, None, 1000 * k, 1500)
        sl = slice(1000 * k, 1000 * k + 1500)
        loss = ppo_loss(ref, obs[sl], act[sl], logp[sl], adv[sl], ret[sl], agent.clip, agent.ent_coef)[0]
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
    assert agent.adam_t == 3
    assert torch.allclose(agent.flat, ref.flat_params(), atol=2e-5), (agent.flat - ref.flat_params()).abs().max()
    assert not torch.allclose(agent.flat, MlpPolicy(30, 1, (32, 16)).to(dev).flat_params(), atol=1e-3)
    o = torch.randn(16, 30, device=dev)                        # the module reads the updated weights (views of the flat tensor)
    assert torch.allclose(agent.policy.pi(o), ref.pi(o), atol=1e-4)


@pytest.mark.gpu
def test_gae_kernel_matches_the_loop():
    from pcc_rl_amd.ppo import gae_fused
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    T, N = 37, 1000
    r, v = torch.randn(T, N, device=dev), torch.randn(T, N, device=dev)
    d = torch.rand(T, N, device=dev) < 0.05
    lv = torch.randn(N, device=dev)
    a1, r1 = gae(r, v, d, lv, 0.99, 0.95)
    a2, r2 = gae_fused(r, v, d, lv, 0.99, 0.95)
    assert torch.allclose(a1, a2, atol=1e-5) and torch.allclose(r1, r2, atol=1e-5)


@pytest.mark.gpu
def test_grouped_rollout_is_the_same_rollout():
    """PPO.collect() over a GroupedNetworkEnv (the same envs as two groups on their own streams, policy kernel and env step
    of a group queued on its stream, no join inside the rollout: double-buffered sampling) gathers exactly what it gathers
    over one batch: same observations, actions, log-probabilities, rewards, advantages -- and the update after it leaves the
    same weights."""
    import pcc_rl_amd
    from pcc_rl_amd.ppo import PPO
    N, T = 2048, 12
    out = []
    for grouped in (False, True):
        env = (pcc_rl_amd.GroupedNetworkEnv(N, 2, device="cuda:0", seed=3) if grouped
               else pcc_rl_amd.BatchedNetworkEnv(N, device="cuda:0", seed=3))
        agent = PPO(env, horizon=T, seed=1, minibatch=N * T // 2, epochs=1)
        obs_b, act_b, logp_b, adv, ret, rew = agent.collect()
        agent.update(obs_b, act_b, logp_b, adv, ret)
        torch.cuda.synchronize()
        out.append([t.clone() for t in (obs_b, act_b, logp_b, adv, ret, rew, agent.policy.flat_params())])
        env.close()
    for a, b, name in zip(out[0], out[1], ("obs", "actions", "logp", "advantages", "returns", "rewards", "weights after the update")):
        assert torch.equal(a, b), name
