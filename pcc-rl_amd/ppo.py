"""On-device PPO for the batched env.

The counterpart of the reference's training script (src/gym/stable_solve.py:39-58): the same
policy shape (separate pi / vf MLPs, hidden sizes --arch = 32,16, tanh), gamma 0.99, a constant
learning-rate schedule and an optimiser minibatch of 2048, with PPO1's other defaults (clip 0.2,
entropy coefficient 0.01, 4 epochs, step size 1e-3, lambda 0.95).  The reference collects 8192
steps from ONE env per iteration; here an iteration is T steps of N envs, everything -- rollout
buffers, advantage estimation, updates -- lives on the GPU and the env never leaves it.

This is a caller of the hot path, not part of it: stable-baselines is not available in this
image, so there is nothing to check agent-level parity against.  What is
checked (tests/test_ppo.py): the GAE recursion and the clipped objective against plain loops (CPU
tests); on the GPU the HIP kernels of the fused path -- policy forward (pcc_policy.hip), the
fp32-MFMA gradient kernel and the Adam step (pcc_ppo.hip: pcc_ppo_minibatch_step), the GAE kernel
-- against float64 autograd, torch.optim.Adam and the loop; that a short run on the GPU improves
the return; and what the whole loop costs next to the env alone (tools/ppo_throughput.py,
profiles/r04_v2_ppo_throughput.json).
"""
import math

import torch
from torch import nn


def mlp(inp, hidden, out):
    layers, last = [], inp
    for h in hidden:
        layers += [nn.Linear(last, h), nn.Tanh()]
        last = h
    layers.append(nn.Linear(last, out))
    return nn.Sequential(*layers)


_warned = set()


def _warn_once(msg):
    """A fallback off the HIP kernels of this caller is never silent (once per message and process)."""
    if msg not in _warned:
        _warned.add(msg)
        import warnings
        warnings.warn("pcc_rl_amd.ppo: " + msg, RuntimeWarning, stacklevel=3)


class MlpPolicy(nn.Module):
    """pi and vf networks of src/gym/stable_solve.py:39-45 (net_arch = [dict(pi=arch, vf=arch)])
    with a state-independent log-std, as stable-baselines' diagonal Gaussian head."""

    def __init__(self, obs_dim, act_dim=1, arch=(32, 16)):
        super().__init__()
        self.pi = mlp(obs_dim, arch, act_dim)
        self.vf = mlp(obs_dim, arch, 1)
        self.log_std = nn.Parameter(torch.zeros(act_dim))

    def dist(self, obs):
        return torch.distributions.Normal(self.pi(obs), self.log_std.exp())

    def value(self, obs):
        return self.vf(obs).squeeze(-1)

    @torch.no_grad()
    def act(self, obs, stochastic=True):
        d = self.dist(obs)
        a = d.sample() if stochastic else d.mean
        return a, d.log_prob(a).sum(-1), self.value(obs)

    def flat_params(self):
        """The parameter block pcc_policy_act reads (include/pcc_policy.h): pi {W1, b1, W2, b2, W3, b3, log_std}, vf {...}."""
        def net(seq):
            return [p.detach().reshape(-1) for m in seq if isinstance(m, nn.Linear) for p in (m.weight, m.bias)]
        return torch.cat(net(self.pi) + [self.log_std.detach().reshape(-1)] + net(self.vf)).float().contiguous()

    def share_flat(self):
        """Move every parameter into ONE flat fp32 tensor in flat_params() order -- the module's parameters become views of
        it -- and return it: the fused optimiser step (pcc_ppo_minibatch_step) then updates the weights the framework
        path and pcc_policy_act read, with no copy in either direction."""
        flat = self.flat_params().clone()
        off = 0

        def net(seq):
            return [p for m in seq if isinstance(m, nn.Linear) for p in (m.weight, m.bias)]
        for prm in net(self.pi) + [self.log_std] + net(self.vf):
            n = prm.numel()
            prm.data = flat[off:off + n].view(prm.shape)
            off += n
        assert off == flat.numel()
        return flat

    def fused_ok(self, obs):
        """Whether pcc_policy_act covers this policy and observation batch (two hidden layers, one action, fp32 on the GPU)."""
        linears = [m for m in self.pi if isinstance(m, nn.Linear)]
        return len(linears) == 3 and linears[2].out_features == 1 and obs.is_cuda and obs.dtype == torch.float32

    @torch.no_grad()
    def act_fused(self, obs, stochastic=True, params=None, noise=None, out=None):
        """act() as ONE kernel launch of the HIP library; returns (action [N, 1], log-probability [N], value [N]) like act().
        A rollout loop passes `params` (flat_params(), built once per rollout -- it is a 13-tensor torch.cat), its own
        `noise` row and `out` = (action, logp, value) rows of its buffers, so that a step adds no framework launch."""
        import ctypes

        from .native import lib
        if not self.fused_ok(obs):
            _warn_once("the policy forward runs on the framework path (no pcc_policy_act for this policy / observation batch)")
            return self.act(obs, stochastic)
        linears = [m for m in self.pi if isinstance(m, nn.Linear)]
        n, D = obs.shape
        if params is None:
            params = self.flat_params()
        if noise is None and stochastic:
            noise = torch.randn(n, device=obs.device)
        if out is None:
            a = torch.empty(n, device=obs.device)
            logp, v = torch.empty_like(a), torch.empty_like(a)
        else:
            a, logp, v = out
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        rc = lib().pcc_policy_act(ptr(obs.contiguous()), n, D, ptr(params), linears[0].out_features, linears[1].out_features,
                                  ptr(noise if stochastic else None), None, ptr(a), ptr(logp), ptr(v),
                                  ctypes.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
        if rc != 0:   # e.g. an observation length without a kernel instantiation
            _warn_once("pcc_policy_act has no kernel for %d observations x hidden %d-%d: the policy forward runs on the framework "
                       "path (several times slower)" % (D, linears[0].out_features, linears[1].out_features))
            a2, logp2, v2 = self.act(obs, stochastic)
            if out is not None:
                a.copy_(a2.reshape(-1)); logp.copy_(logp2); v.copy_(v2)
            return a2, logp2, v2
        return a.reshape(n, 1), logp, v


def gae(rewards, values, dones, last_value, gamma=0.99, lam=0.95):
    """Generalised advantage estimation over [T, N] tensors.  dones[t] marks that the env was reset
    after step t (the next observation belongs to a new episode)."""
    T = rewards.shape[0]
    adv = torch.zeros_like(rewards)
    running = torch.zeros_like(last_value)
    next_value = last_value
    for t in range(T - 1, -1, -1):
        alive = 1.0 - dones[t].to(rewards.dtype)
        delta = rewards[t] + gamma * next_value * alive - values[t]
        running = delta + gamma * lam * alive * running
        adv[t] = running
        next_value = values[t]
    return adv, adv + values


def gae_fused(rewards, values, dones, last_value, gamma=0.99, lam=0.95):
    """gae() as one launch of the HIP library (pcc_gae: thread = env, T steps backwards) for fp32 [T, N] rows on the GPU."""
    import ctypes

    from .native import lib
    if not (rewards.is_cuda and rewards.dtype == torch.float32 and rewards.dim() == 2):
        return gae(rewards, values, dones, last_value, gamma, lam)
    T, N = rewards.shape
    rewards, values, last_value = rewards.contiguous(), values.contiguous(), last_value.contiguous().float()
    d8 = dones.contiguous().view(torch.uint8) if dones.dtype == torch.bool else dones.to(torch.uint8).contiguous()
    adv, ret = torch.empty_like(rewards), torch.empty_like(rewards)
    ptr = lambda t: ctypes.c_void_p(t.data_ptr())
    rc = lib().pcc_gae(ptr(rewards), ptr(values), ptr(d8), ptr(last_value), T, N, gamma, lam, ptr(adv), ptr(ret),
                       ctypes.c_void_p(torch.cuda.current_stream(rewards.device).cuda_stream))
    if rc != 0:
        raise RuntimeError("pcc_gae failed (%d)" % rc)
    return adv, ret


def ppo_loss(policy, obs, act, logp_old, adv, ret, clip=0.2, ent_coef=0.01):
    """PPO1's objective on one minibatch: clipped surrogate + 0.5 * value error - ent_coef * entropy.
    Returns (loss, policy term, value term, entropy)."""
    d = policy.dist(obs)
    logp = d.log_prob(act).sum(-1)
    ratio = (logp - logp_old).exp()
    pg = -torch.min(ratio * adv, ratio.clamp(1 - clip, 1 + clip) * adv).mean()
    vf = 0.5 * (policy.value(obs) - ret).pow(2).mean()
    ent = d.entropy().sum(-1).mean()
    return pg + vf - ent_coef * ent, pg, vf, ent


class PPO(object):
    def __init__(self, env, arch=(32, 16), gamma=0.99, lam=0.95, clip=0.2, ent_coef=0.01, lr=1e-3,
                 epochs=4, minibatch=None, horizon=64, seed=0, fused_update=True):
        """minibatch None = a quarter of the rollout, at least 2048: the reference's ratio (optim_batchsize 2048 of a
        timesteps_per_actorbatch of 8192, stable_solve.py:52) -- at 65 536 envs x 64 steps a fixed 2048 would be 2 048
        optimiser steps per epoch, thousands of launches of a few microseconds of work each."""
        self.env, self.gamma, self.lam, self.clip, self.ent_coef = env, gamma, lam, clip, ent_coef
        self.epochs, self.minibatch, self.horizon = epochs, minibatch, horizon
        torch.manual_seed(seed)
        if hasattr(env, "groups"):   # GroupedNetworkEnv: the surface of a BatchedNetworkEnv, from its groups
            env.obs_dim, env.n_senders = env.groups[0].obs_dim, env.groups[0].n_senders
        self.policy = MlpPolicy(env.obs_dim, 1, arch).to(env.device)
        if hasattr(env, "groups") and not (len(arch) == 2 and env.n_senders == 1 and torch.device(env.device).type == "cuda"):
            # (a GroupedNetworkEnv is stepped group by group on its streams by the fused rollout only: it has no step() of its own)
            raise ValueError("PPO over a GroupedNetworkEnv needs the fused rollout (a two-hidden-layer policy, one sender, on the GPU)")
        self.lr, self.adam_eps = lr, 1e-5
        # the fused optimiser step (pcc_ppo_minibatch_step: gradient + Adam in two launches) when the library has a kernel
        # for this shape; the framework path (autograd + torch.optim.Adam, the same arithmetic) otherwise
        self.fused_update = bool(fused_update) and self._fused_update_ok()
        if fused_update and not self.fused_update:
            _warn_once("PPO.update runs autograd + torch.optim.Adam (no pcc_ppo_minibatch_step for observation length %d, hidden %s, "
                       "%d sender(s) on %s): about 20x slower than the fused step" % (env.obs_dim, list(arch), env.n_senders, env.device))
        if self.fused_update:
            from .native import lib
            self.flat = self.policy.share_flat()
            self.adam_m, self.adam_v, self.adam_t = torch.zeros_like(self.flat), torch.zeros_like(self.flat), 0
            self.scratch = torch.empty(lib().pcc_ppo_scratch_floats(env.obs_dim, arch[0], arch[1]), device=env.device)
            self.stats_buf = torch.zeros(4, device=env.device)
        self.opt = torch.optim.Adam(self.policy.parameters(), lr=lr, eps=self.adam_eps)
        self.obs = env.reset().clone()
        if self.minibatch is None:
            self.minibatch = max(2048, env.n_envs * horizon // 4)

    def _fused_update_ok(self):
        env = self.env
        arch = [m.out_features for m in self.policy.pi if isinstance(m, nn.Linear)]
        return (torch.device(env.device).type == "cuda" and arch == [32, 16, 1] and env.obs_dim in (30, 12, 6, 3)
                and env.n_senders == 1)

    def collect(self):
        """One rollout of `horizon` steps of every env.  The policy kernel reads the observation row the env wrote and
        writes action / log-probability / value into the rollout rows; the env reads that action row and writes the next
        observation, reward and done rows: two library calls and three kernels per step, no framework launch, no copy."""
        env, T, N = self.env, self.horizon, self.env.n_envs
        dev = env.device
        obs_b = torch.empty((T + 1, N, env.obs_dim), device=dev)   # row t: what the policy saw at step t; row T: the last
        act_b = torch.empty((T, N, 1), device=dev)
        logp_b = torch.empty((T, N), device=dev)
        val_b = torch.empty((T, N), device=dev)
        rew_b = torch.empty((T, N), device=dev)
        done_b = torch.empty((T, N), dtype=torch.bool, device=dev)
        obs_b[0] = self.obs
        fused = self.policy.fused_ok(obs_b[0]) and env.n_senders == 1
        groups = getattr(env, "groups", None)
        if fused and groups is not None:
            # Double-buffered sampling (GroupedNetworkEnv: the same envs as G groups on their own streams): group g's
            # policy kernel and env step are queued on stream g, nothing joins the groups inside the rollout -- while one
            # group's env launches run out their tails (a third of the wavefront slots busy), the
            # other group's policy kernel and launches fill the machine.  Same numbers as one batch: a group holds the
            # global env ids g * n .. and reads its own rows of every buffer.
            params = self.policy.flat_params()
            noise = torch.randn((T, N), device=dev)
            n = env.group_size
            cur = torch.cuda.current_stream(dev)
            for s in env.streams:
                s.wait_stream(cur)                                 # the buffers and the noise were made on this stream
            for t in range(T):
                for g, eg in enumerate(groups):
                    lo, hi = g * n, (g + 1) * n
                    with torch.cuda.stream(env.streams[g]):
                        self.policy.act_fused(obs_b[t, lo:hi], True, params, noise[t, lo:hi],
                                              (act_b[t, lo:hi].reshape(n), logp_b[t, lo:hi], val_b[t, lo:hi]))
                        eg.step_into(act_b[t, lo:hi], obs_b[t + 1, lo:hi], rew_b[t, lo:hi], done_b[t, lo:hi])
            for s in env.streams:
                cur.wait_stream(s)
        elif fused:
            params = self.policy.flat_params()                     # once per rollout, not per step
            noise = torch.randn((T, N), device=dev)                # the horizon's draws in one launch
            for t in range(T):
                self.policy.act_fused(obs_b[t], True, params, noise[t], (act_b[t].reshape(N), logp_b[t], val_b[t]))
                env.step_into(act_b[t], obs_b[t + 1], rew_b[t], done_b[t])   # tensors in, tensors out, no host round trip
        else:
            for t in range(T):
                a, logp, v = self.policy.act(obs_b[t])
                act_b[t], logp_b[t], val_b[t] = a, logp, v
                nobs, r, d, _ = env.step(a)
                rew_b[t], done_b[t], obs_b[t + 1] = r, d, nobs
        obs = obs_b[T]
        self.obs = obs.clone()
        obs_b = obs_b[:T]
        with torch.no_grad():
            last_v = self.policy.value(obs)
        # never train on corrupted rollouts: an overflowed in-flight ring / an empty ring pool (a trained
        # policy can push many deep-queue envs to MAX_RATE: BatchedNetworkEnv(ring_pools=...)) is flagged, not silent
        env.check_flags()
        adv, ret = (gae_fused if fused else gae)(rew_b, val_b, done_b, last_v, self.gamma, self.lam)
        return obs_b, act_b, logp_b, adv, ret, rew_b

    def update(self, obs_b, act_b, logp_b, adv, ret):
        n = obs_b.shape[0] * obs_b.shape[1]
        obs_f, act_f = obs_b.reshape(n, -1), act_b.reshape(n, -1)
        logp_f, adv_f, ret_f = logp_b.reshape(n), adv.reshape(n), ret.reshape(n)
        adv_f = (adv_f - adv_f.mean()) / (adv_f.std() + 1e-8)
        if self.fused_update:
            return self._update_fused(obs_f, act_f, logp_f, adv_f, ret_f)
        stats = {}
        for _ in range(self.epochs):
            perm = torch.randperm(n, device=obs_f.device)
            for i in range(0, n, self.minibatch):
                idx = perm[i:i + self.minibatch]
                loss, pg, vf, ent = ppo_loss(self.policy, obs_f[idx], act_f[idx], logp_f[idx], adv_f[idx], ret_f[idx],
                                             self.clip, self.ent_coef)
                self.opt.zero_grad(set_to_none=True)
                loss.backward()
                self.opt.step()
                stats = {"pg": pg.detach(), "vf": vf.detach(), "entropy": ent.detach()}
        return {k: float(v) for k, v in stats.items()}

    def minibatch_step_fused(self, obs_f, act_f, logp_f, adv_f, ret_f, perm, start, count, lr=None, grad_out=None):
        """One optimiser step on samples perm[start : start + count] of the flattened rollout as two launches of the HIP
        library (include/pcc_policy.h: pcc_ppo_minibatch_step).  lr=0: gradient only (into grad_out)."""
        import ctypes

        from .native import lib
        lr = self.lr if lr is None else lr
        if start < 0 or count < 1 or start + count > (perm.numel() if perm is not None else obs_f.shape[0]):
            raise ValueError("minibatch [%d, %d) outside the rollout" % (start, start + count))
        if lr != 0.0:
            self.adam_t += 1
        ptr = lambda t: None if t is None else ctypes.c_void_p(t.data_ptr())
        D = obs_f.shape[1]
        rc = lib().pcc_ppo_minibatch_step(ptr(obs_f), ptr(act_f), ptr(logp_f), ptr(adv_f), ptr(ret_f), ptr(perm), start, count,
                                          D, 32, 16, ptr(self.flat), ptr(self.adam_m), ptr(self.adam_v), max(self.adam_t, 1),
                                          lr, 0.9, 0.999, self.adam_eps, self.clip, self.ent_coef, ptr(self.scratch),
                                          ptr(grad_out), ptr(self.stats_buf),
                                          ctypes.c_void_p(torch.cuda.current_stream(obs_f.device).cuda_stream))
        if rc != 0:
            raise RuntimeError("pcc_ppo_minibatch_step failed (%d)" % rc)

    def _update_fused(self, obs_f, act_f, logp_f, adv_f, ret_f):
        n = obs_f.shape[0]
        obs_f, act_f = obs_f.contiguous(), act_f.reshape(n).contiguous()
        logp_f, adv_f, ret_f = logp_f.contiguous(), adv_f.contiguous(), ret_f.contiguous()
        for _ in range(self.epochs):
            perm = torch.randperm(n, device=obs_f.device)
            for i in range(0, n, self.minibatch):
                self.minibatch_step_fused(obs_f, act_f, logp_f, adv_f, ret_f, perm, i, min(self.minibatch, n - i))
        st = self.stats_buf.tolist()   # of the last minibatch, like the framework path
        ent = float(self.policy.log_std.detach().sum()) + 0.5 * (1.0 + math.log(2.0 * math.pi)) * self.policy.log_std.numel()
        return {"pg": -st[0], "vf": 0.5 * st[1], "entropy": ent, "clip_frac": st[2]}

    def iterate(self):
        obs_b, act_b, logp_b, adv, ret, rew = self.collect()
        stats = self.update(obs_b, act_b, logp_b, adv, ret)
        stats["mean_step_reward"] = float(rew.mean())
        return stats
